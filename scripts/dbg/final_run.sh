# the round's closing measurements (one box): full GPU suite, smoke, bench (both modes), training steps, variant A/Bs
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/final/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/final/gpu_tests.log
cp gpurun_out/tol_stats.json gpurun_out/final/tol_stats.json 2>/dev/null
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/final/smoke.log
timeout -k 10 400 python bench.py > gpurun_out/final/bench_n1.json 2> gpurun_out/final/bench_n1.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --mode step > gpurun_out/final/bench_step_n1.json 2> gpurun_out/final/bench_step.err; echo "step rc=$?"
WARM=4 timeout -k 10 600 python scripts/bench_train.py --kinds gcn,sage,gin > gpurun_out/final/train.jsonl 2> gpurun_out/final/train.err; echo "train rc=$?"
WARM=4 timeout -k 10 300 python scripts/bench_train.py --kinds idgin,idgcn --steps 20 >> gpurun_out/final/train.jsonl 2>> gpurun_out/final/train.err; echo "train-id rc=$?"
rm -f gpurun_out/final/fused_variants.jsonl
for cfg in "DIM=256" "DIM=512" "DIM=256 DOUT=512" "DIM=256 SELF=1" "DIM=512 SELF=1" "DIM=128"; do env $cfg VARIANTS=0,1,9 timeout -k 10 200 python scripts/dbg/fused_variants.py 2>/dev/null >> gpurun_out/final/fused_variants.jsonl; done; echo variants done
for V in 1 0; do echo "MP_WGRAD_PC=$V"; MP_WGRAD_PC=$V timeout -k 10 300 python scripts/dbg/wgrad_bench.py 2>&1 | grep "^M="; done > gpurun_out/final/wgrad.txt; echo wgrad done
for V in 1 0; do for D in 256 512; do DIM=$D MP_X3_PC=$V timeout -k 10 100 python scripts/dbg/x3_time.py 2>/dev/null; done; done > gpurun_out/final/x3.txt; echo x3 done
