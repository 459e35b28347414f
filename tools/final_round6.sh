# The builder-run differential fuzz / stress evidence of round 6 on the final library (after the api.hip split and the tws_acquire fix):
#   bash tools/final_round6.sh > gpurun_out/r06_fuzz_final.txt
cd "$(dirname "$0")/.."
echo "== tools/fuzz_msm.sh (CASES=40 SEED=83): window layouts x plain / streamed chunks, table mode, 2 / 3 / 8 logical devices, one-lane tails"
CASES=40 SEED=83 bash tools/fuzz_msm.sh 2>&1 | grep -E "^fuzz|MISMATCH"
echo "== tools/fuzz_ntt.py: default, barrier kernel without folds, 32-MiB table budget"
python tools/fuzz_ntt.py --cases 300 --seed 83 --max-log 22 2>&1 | tail -2
MI355ZK_NTT_WAVELOCAL=0 MI355ZK_NTT_NO_FOLD=1 python tools/fuzz_ntt.py --cases 200 --seed 84 --max-log 21 2>&1 | tail -2
MI355ZK_NTT_TABLES_GB=0.03 python tools/fuzz_ntt.py --cases 200 --seed 85 --max-log 21 2>&1 | tail -2
echo "== tools/fuzz_rows.py: one and three logical devices"
python tools/fuzz_rows.py --cases 120 --seed 83 2>&1 | tail -2
python tools/fuzz_rows.py --cases 60 --seed 84 --devices 3 2>&1 | tail -2
echo "== tools/stress_threads.py: 8 threads x 40 s, one and three logical devices, with the profiling events on"
python tools/stress_threads.py --threads 8 --seconds 40 --seed 83 2>&1 | tail -3
python tools/stress_threads.py --threads 8 --seconds 30 --seed 84 --devices 3 --prof 2>&1 | tail -3
echo "== the same stress under ThreadSanitizer (tools/run_tsan.sh), 20 s: reports naming this library"
rm -f gpurun_out/tsanstress*
TSAN_LOG=$PWD/gpurun_out/tsanstress tools/run_tsan.sh python tools/stress_threads.py --threads 8 --seconds 20 --seed 85 2>&1 | tail -2
grep -l "libmi355zk" gpurun_out/tsanstress* 2>/dev/null | wc -l
grep -h "^SUMMARY" gpurun_out/tsanstress* 2>/dev/null | sort | uniq -c | sort -rn | head -5
