# closing evidence of the round on one box: tests, smoke, bench (+ rocprofv3 passes of the same command), steps, A/Bs
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/final/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/final/gpu_tests.log
cp gpurun_out/tol_stats.json gpurun_out/final/tol_stats.json 2>/dev/null
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 400 python bench.py > gpurun_out/final/bench_n1.json 2> gpurun_out/final/bench_n1.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --mode step > gpurun_out/final/bench_step_n1.json 2> gpurun_out/final/bench_step.err; echo "step rc=$?"
bash scripts/prof_run.sh r03 > gpurun_out/final/prof_run.log 2>&1; echo "prof_run rc=$?"
python3 scripts/summarize_prof.py gpurun_out/prof_r03 r03 gcn_norm_sum_d256_BA_n10000000_m5 > gpurun_out/final/summarize.log 2>&1; echo "summarize rc=$?"
mkdir -p gpurun_out/final/profiles && cp profiles/r03_kernel_stats.csv profiles/r03_pmc.json profiles/pmc_traffic.json gpurun_out/final/profiles/
find gpurun_out/prof_r03 -type f -size +1M -delete
WARM=4 timeout -k 10 600 python scripts/bench_train.py --kinds gcn,sage,gin > gpurun_out/final/train.jsonl 2> gpurun_out/final/train.err; echo "train rc=$?"
WARM=4 timeout -k 10 300 python scripts/bench_train.py --kinds idgin,idgcn --steps 20 >> gpurun_out/final/train.jsonl 2>> gpurun_out/final/train.err; echo "train-id rc=$?"
timeout -k 10 500 python scripts/kernel_table.py > gpurun_out/final/kernel_table.log 2>&1; echo "table rc=$?"
for T in 1 0; do MP_AGG_TILES=$T timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --d 512 2>/dev/null | grep "^{" > gpurun_out/final/bench_d512_tiles$T.json; done; echo d512 done
