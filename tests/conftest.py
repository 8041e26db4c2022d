import os
import sys

# the decoders' pre-decode kernels are for batches (KMP_PRE_MIN_BATCH, default 256 entries); the suite's small cases run them too
os.environ.setdefault("KMP_PRE_MIN_BATCH", "1")
# the bulk engines of large host-memory batches: pieces of 2 048 slices here (default 16 384: 4 GiB of pinned staging each)
os.environ.setdefault("KMP_HOST_BULK_SLICES", "2048")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
