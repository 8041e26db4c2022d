"""Python mirror of the reference's core contract, kept field-for-field so the
parity tests read like the reference's own:

  ByteArraySlice   kompressor-core/src/commonMain/kotlin/com/ensody/kompressor/core/ByteArraySlice.kt:14-90
  SliceTransform   .../core/SliceTransform.kt:8-30
  transform(bytes) .../core/SliceTransform.kt:33-56  (one-shot driver)
"""


class ByteArraySlice:
    """Readable region [read_start, write_start), writable region [write_start, write_limit)."""

    def __init__(self, data, read_start=0, write_start=None, write_limit=None, insufficient=False):
        if isinstance(data, int):                       # ByteArraySlice(size): empty, ready to be filled
            data = bytearray(data)
            write_start = 0 if write_start is None else write_start
        elif not isinstance(data, bytearray):
            data = bytearray(data)
        self.data = data
        self.read_start = read_start
        self.write_start = len(data) if write_start is None else write_start
        self.write_limit = len(data) if write_limit is None else write_limit
        self.insufficient = insufficient

    @property
    def remaining_read(self):
        return self.write_start - self.read_start

    @property
    def remaining_write(self):
        return self.write_limit - self.write_start

    @property
    def has_data(self):
        return self.remaining_read != 0

    @property
    def is_full(self):
        return self.write_start == self.write_limit

    def __repr__(self):
        return f"ByteArraySlice(remainingRead={self.remaining_read}, insufficient={self.insufficient})"


class SliceTransform:
    """transform(input, output, finish): consume input[read_start:write_start], produce into
    output[write_start:write_limit], advance both cursors, set output.insufficient when more
    room is needed (SliceTransform.kt:8-30)."""

    def transform(self, input, output, finish):          # noqa: A002 - reference's argument names
        raise NotImplementedError

    def transform_bytes(self, data: bytes) -> bytes:
        """SliceTransform.transform(ByteArray) (SliceTransform.kt:33-45)."""
        input_slice = ByteArraySlice(bytearray(data))
        block_size = max(8192, len(data) // 10)
        output_slices = [ByteArraySlice(block_size)]
        while True:
            output_slice = output_slices[-1]
            self.transform(input_slice, output_slice, finish=True)
            if output_slice.insufficient:
                output_slices.append(ByteArraySlice(block_size))
            if not (input_slice.remaining_read != 0 or output_slice.insufficient):
                break
        return b"".join(bytes(s.data[s.read_start:s.write_start]) for s in output_slices)   # getOutput, :47-56
