#!/usr/bin/env python3
"""Turns gpurun_out/prof/ (written on the GPU box by tools/profile_round.sh: rocprofv3 result databases + bench lines)
into profiles/<name>.txt and profiles/pmc_latest.json.

    python tools/summarize_profiles.py r01_final
"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "prof")


def top(db, limit=6):
    rows = sqlite3.connect(db).execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    out = ["Name,Calls,TotalDurationUs,AverageUs,Percentage"]
    for name, calls, tot, avg, pct in rows[:limit]:
        if len(name) > 90:
            name = name[:87] + "..."
        out.append(f"{name},{calls},{tot:.1f},{avg:.1f},{pct:.2f}")
    return out


def counters(db):
    rows = sqlite3.connect(db).execute(
        "select kernel_name,counter_name,sum(value),count(*) from counters_collection group by kernel_name,counter_name").fetchall()
    return [(k, c, v, n) for k, c, v, n in rows if k.startswith("void k_") or k.startswith("k_")]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "r02_final"
    # on the GPU box the result databases are too large to travel back: profile_round.sh runs this script there with an
    # output directory under gpurun_out/ and deletes the databases; here the two files are then copied into profiles/
    outdir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")
    os.makedirs(outdir, exist_ok=True)
    lines = [f"# {name}: MI355X, 1 GPU; collected on the GPU box by tools/profile_round.sh, summarised by tools/summarize_profiles.py.",
             "# Durations in microseconds (rocprofv3 --kernel-trace --stats, result database view top_kernels).", ""]
    runs = [("compress", "python3 bench.py --steps 5 --warmup 1 --no-stream     (compress, BASELINE configs[1]; the figures of the streaming entry point are in the bench lines)"),
            ("config3", "python3 bench.py --config 3 --steps 3 --warmup 1 --no-cpu --no-stream   (BASELINE configs[3]'s share of one GPU: 131 072 text / binary slices, two launches of each kernel per step)"),
            ("decompress", "python3 bench.py --mode decompress --steps 3 --warmup 1 --no-cpu   (configs[2]; the frames are compressed first)"),
            ("deflate", "python3 bench.py --mode deflate --steps 1 --warmup 0   (configs[4]: four pieces of 16 384 slices, search of one beside parse + encode of the previous)"),
            ("big1m", "python3 bench.py --slice-kib 1024 --slices 8192 --steps 2 --warmup 1   (north_star slice-size sweep: 1 MiB slices, frames of several blocks)"),
            ("big256k", "python3 bench.py --slice-kib 256 --slices 32768 --steps 2 --warmup 1 --no-cpu   (slice-size sweep: 256 KiB slices)"),
            ("level1", "python3 bench.py --level 1 --steps 3 --warmup 1 --no-cpu   (65 536 x 64 KiB at level 1, the Ktor encoder's level)"),
            ("level7", "python3 bench.py --level 7 --slices 16384 --steps 2 --warmup 1 --no-cpu   (16 384 x 64 KiB at level 7: libzstd's lazy2 parse over its row-based match finder, zstd_lazy.h)"),
            ("dict_trained", "python3 bench.py --dict-kib 64 --dict-trained --slice-kib 8 --slices 262144 --steps 3 --warmup 1 --no-cpu   (8 KiB records with a 64 KiB dictionary trained by the box's ZDICT)"),
            ("deflate1", "python3 bench.py --mode deflate --deflate-level 1 --steps 1 --warmup 0 --no-cpu   (raw DEFLATE level 1 = deflate_fast: one k_deflate_fast launch over the batch, then the shared encoder)"),
            ("deflate256k", "python3 bench.py --mode deflate --slice-kib 256 --slices 8192 --steps 2 --warmup 1 --no-cpu   (raw DEFLATE level 6 on 256 KiB slices: the sort + parse kernels in 64 KiB spans, 7 + 7 launches a piece)"),
            ("deflate_w12m5", "python3 bench.py --mode deflate --deflate-window-bits 12 --deflate-mem-level 5 --slices 16384 --steps 2 --warmup 1 --no-cpu   (deflateInit2's windowBits 12, memLevel 5)"),
            ("inflate", "python3 bench.py --mode inflate --steps 3 --warmup 1 --no-cpu   (ZlibDecompressor over the 65 536 level-6 streams of configs[4]; the streams are made first)")]
    for key, cmd in runs:
        db = os.path.join(P, key, "run_results.db")
        if not os.path.exists(db):
            continue
        lines.append(f"## rocprofv3 --kernel-trace --stats -- {cmd}")
        lines += top(db)
        lines.append("bench line of the same run:")
        lines.append(open(os.path.join(P, key + ".json")).read().strip())
        lines.append("")
    lines.append("## PMC passes (each its own run: rocprofv3 --pmc <counters> -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-pcie --no-stream: every launch of a kernel in such a pass is one full batch); sums over the launches of a kernel")
    allc = {}
    for d in sorted(os.listdir(P)):
        db = os.path.join(P, d, "run_results.db")
        if d.startswith("pmc_") and os.path.exists(db):
            for k, c, v, n in counters(db):
                allc[(k, c)] = (v, n)
                lines.append(f"{k:40s} {c:24s} {v:.6g}   ({n} launches)")
    lines.append("")
    lines.append("## HBM bytes of the other kernels (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, own runs: bench.py --mode decompress --steps 1 --warmup 0; --mode deflate --slices 16384 --steps 1 --warmup 0)")
    other = {}
    for d in sorted(os.listdir(P)):
        db = os.path.join(P, d, "run_results.db")
        if d.startswith("pmc2_") and os.path.exists(db):
            for k, c, v, n in counters(db):
                if "k_zstd_decode" in k or "predecode" in k or "k_deflate" in k or "k_inflate" in k:
                    other[(k.split("(")[0], c)] = (v, n)
                    lines.append(f"{k:40s} {c:24s} {v:.6g}   ({n} launches)")
    lines.append("")
    lines.append("## HBM bytes of the other parsers (own runs, bench.py --steps 1 --warmup 0 --no-cpu + --level 1 | --dict-kib 16 | --slice-kib 256 --slices 32768 | --slice-kib 1024 --slices 8192)")
    third = {}
    for d in sorted(os.listdir(P)):
        db = os.path.join(P, d, "run_results.db")
        if d.startswith("pmc3_") and os.path.exists(db):
            tag = d[len("pmc3_"):].rsplit("_", 2)[0]
            for k, c, v, n in counters(db):
                if any(x in k for x in ("k_zstd_match_fast", "k_zstd_match_dict", "k_zstd_big", "k_zstd_entropy")):
                    third[(tag, k.split("(")[0].replace("void ", "").split("<")[0], c)] = (v, n)
                    lines.append(f"[{tag}] {k:50s} {c:14s} {v:.6g}   ({n} launches)")
    open(os.path.join(outdir, name + ".txt"), "w").write("\n".join(lines) + "\n")
    mk = [k for (k, c) in allc if "k_zstd_match" in k and c == "FETCH_SIZE"]
    if mk:
        k = mk[0]
        fetch, nl = allc[(k, "FETCH_SIZE")]
        write, _ = allc[(k, "WRITE_SIZE")]
        line = json.loads(open(os.path.join(P, "compress.json")).read())
        pj = {"source": f"profiles/{name}.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, bench.py --steps 1 --warmup 0 --no-pcie --no-stream: every counted launch is a full batch)",
              "slices": line["config"]["slices_per_gpu"], "team": line["config"]["team_lanes"],
              "launches": line["roofline"].get("launches_per_step", 2), "launches_in_the_counter_pass": nl,
              "zstd_match_fetch_kib": fetch, "zstd_match_write_kib": write,
              "zstd_match_hbm_bytes_per_launch": int((fetch + write) * 1024 / nl),
              "zstd_match_read_requests_per_launch": int(allc.get((k, "TCC_EA0_RDREQ_sum"), (0, 1))[0] / nl),
              "zstd_match_write_requests_per_launch": int(allc.get((k, "TCC_EA0_WRREQ_sum"), (0, 1))[0] / nl),
              "note": "(FETCH_SIZE+WRITE_SIZE)*1024 / launches; the guide's x2 correction for wide coalesced reads is not applied: "
                      "this kernel's reads are scattered 4- and 8-byte probes (TCC_EA0_RDREQ_32B = 0, RDREQ*64 = FETCH_SIZE)"}
        for kern in ("k_zstd_decode", "k_zstd_seq_predecode", "k_zstd_lit_predecode", "k_deflate_sort", "k_deflate_lazy", "k_deflate_chains", "k_deflate_best", "k_deflate_parse", "k_deflate_encode", "k_inflate", "k_inflate_predecode", "k_inflate_exec"):
            if (kern, "FETCH_SIZE") in other and (kern, "WRITE_SIZE") in other:
                f, nl2 = other[(kern, "FETCH_SIZE")]
                w, _ = other[(kern, "WRITE_SIZE")]
                pj[kern + "_hbm_bytes_per_launch"] = int((f + w) * 1024 / nl2)
        for (tag, kern, ctr) in list(third):
            if ctr == "FETCH_SIZE" and (tag, kern, "WRITE_SIZE") in third:
                f, nl3 = third[(tag, kern, "FETCH_SIZE")]
                w, _ = third[(tag, kern, "WRITE_SIZE")]
                pj[f"{tag}:{kern}_hbm_bytes_per_launch"] = int((f + w) * 1024 / nl3)
        if "k_inflate_predecode_hbm_bytes_per_launch" in pj and "k_inflate_exec_hbm_bytes_per_launch" in pj:
            pj["inflate_pipeline_hbm_bytes_per_step"] = pj["k_inflate_predecode_hbm_bytes_per_launch"] + pj["k_inflate_exec_hbm_bytes_per_launch"]      # (the bench.py --mode inflate pass: 65 536 streams per launch)
        pj["note_other_kernels"] = "k_zstd_decode / k_zstd_seq_predecode / k_zstd_lit_predecode: 65536 frames per launch; k_deflate_*: 16384 slices per launch; k_inflate_predecode / k_inflate_exec: 65536 streams per launch (the --mode inflate pass); (FETCH_SIZE+WRITE_SIZE)*1024 / launches"
        # bench.py compares this with the tree it runs from: counters collected on other kernel sources are flagged, not passed on silently
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        pj["csrc_sha256"] = bench.csrc_sha256()
        json.dump(pj, open(os.path.join(outdir, "pmc_latest.json"), "w"), indent=1)
    print("\n".join(lines[:60]))


if __name__ == "__main__":
    main()
