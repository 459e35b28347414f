#!/usr/bin/env python3
"""Local step after `gpurun -- bash tools/refresh_profiles.sh`: copies the summaries from gpurun_out/refresh/ into
profiles/ (round-tagged names) and rebuilds profiles/latest_pmc.json, which bench.py reports as roofline.traffic."""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "refresh"); DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03_final"
def put(src, dst, header=None):
    body = open(os.path.join(SRC, src)).read()
    with open(os.path.join(DST, f"{tag}_{dst}"), "w") as f:
        if header: f.write(header)
        f.write(body)
put("bench_n1.json", "bench_n1.json"); put("bench_2e20.json", "bench_2e20.json"); put("g2_2e20.json", "g2_2e20.json")
put("next_rows_2e20.json", "next_rows_2e20.json"); put("ntt20.json", "ntt20.json")
put("contribute_2e20.json", "contribute_2e20.json"); put("host_entry.json", "host_entry.json"); put("shard_cells_2e26.json", "shard_cells_2e26.json")
put("ntt_16_20_24.json", "ntt_16_20_24.json"); put("skew_2e26.json", "skew_2e26.json"); put("bench_n1_tau.json", "bench_n1_tau.json")
put("table_mode.json", "table_mode.json"); put("ubench_valu.txt", "ubench_valu.txt"); put("ntt_other_sizes.json", "ntt_other_sizes.json"); put("ubench_fieldmul.txt", "ubench_fieldmul.txt"); put("ubench_wave_bucket.txt", "ubench_wave_bucket.txt"); put("prover.json", "prover.json"); put("skew_small.json", "skew_16_20.json")
put("msm26_kernel_stats.txt", "msm26_kernel_stats.txt",
    "# rocprofv3 --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline   (MI355X, 2^26-point G1 MSM)\n"
    "# summarised from the rocpd database with tools/rocpd_summary.py (ROCm 7.2 rocprofv3 writes rocpd; same numbers as --stats)\n"
    "# batch_exp_kernel = synthetic-input generation (outside the timed region); every msm_* launch is a full-size step\n"
    "# (1 warm-up + 3 timed + 2 of the linearity check); msm_accumulate_kernel is the dominant kernel of a step\n")
put("ntt20_pass_sq_pmc.txt", "ntt20_pass_sq_pmc.txt", "# ntt_pass_kernel, 2^20 elements (2 passes of 1024-point rows, 512 tiles of 2 x 1024, two 512-lane workgroups per CU), per dispatch,\n# rocprofv3 --pmc (two passes of 8 / 7 counters), tools/bench_ntt.py --log-n 20; SQ cycle counters tick once per 4 clocks\n")
put("msm20_timeline.txt", "msm20_timeline.txt", "# rocprofv3 --kernel-trace -- python tools/trace_one_msm.py: the launches of ONE 2^20-point G1 multiexp in order (start offset, duration incl. the\n# profiler's serialisation, gap to the previous kernel); the host join (0.14 ms) follows the last copy\n")
put("ntt20_kernel_stats.txt", "ntt20_kernel_stats.txt", "# rocprofv3 --kernel-trace -- python tools/bench_ntt.py --check   (MI355X, 2^20 Fr NTT, 20 iterations x 4 ops)\n")
put("msm26_accumulate_sq_pmc.txt", "msm26_accumulate_sq_pmc.txt", "# rocprofv3 --pmc SQ_* (one pass, 8 counters) on msm_accumulate_kernel<Fq>, 2^26 points, MI355X\n")
with open(os.path.join(DST, f"{tag}_msm26_pmc_hbm.txt"), "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline  (MI355X)\n"
            "# unit: KB per dispatch.  gfx950 FETCH_SIZE tallies 64 B per L2->fabric request: coalesced 16 B/lane streams (128-byte requests)\n"
            "# read half their bytes (msm_digits_hist_kernel: 2 GiB of scalars -> 1.05e6 KB), random 64-byte records and 16-byte words read\n"
            "# exactly 64 B per access (profiles/r02_ubench_gather_fetch_calibration.txt).  msm_accumulate_kernel is gathers only: no x2.\n")
    f.write(open(os.path.join(SRC, "msm26_pmc_fetch.txt")).read())
    f.write("".join(l for l in open(os.path.join(SRC, "msm26_pmc_write.txt")) if not l.startswith("kernel ")))
def counter(path, kernel):
    for l in open(os.path.join(SRC, path)):
        if l.startswith(kernel + " "): return float(l.split()[3])
    for l in open(os.path.join(SRC, path)):  # (template arguments beyond the field do not survive rocpd_summary's short())
        if l.startswith(kernel.split("<")[0]) and "<Fq>" in l: return float(l.split()[3])
    raise SystemExit(f"{kernel} not in {path}")
fetch = counter("msm26_pmc_fetch.txt", "zk::msm_accumulate_kernel<Fq>"); write = counter("msm26_pmc_write.txt", "zk::msm_accumulate_kernel<Fq>")
sys.path.insert(0, ROOT)
import bench
json.dump({"round": 3, "workload_log_n": 26, "n_gpus": 1, "kernel": "msm_accumulate_kernel<Fq>", "FETCH_SIZE_KB_per_launch": fetch,
           "WRITE_SIZE_KB_per_launch": write, "hbm_bytes_per_launch": int((fetch + write) * 1024), "kernel_sources_sha": bench.kernel_sources_sha(),
           "how": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/{tag}_msm26_pmc_hbm.txt); bytes = (FETCH_SIZE + WRITE_SIZE)*1024: "
                  "the kernel reads through 64-byte base gathers and 16-byte index-list loads, which FETCH_SIZE tallies exactly (one 64-byte "
                  "request each, profiles/r02_ubench_gather_fetch_calibration.txt) -- the guide's x2 correction is for 128-byte streaming requests"},
          open(os.path.join(DST, "latest_pmc.json"), "w"), indent=1)
print(open(os.path.join(DST, "latest_pmc.json")).read())
