# rocprofv3 evidence for the round: kernel trace + FETCH / WRITE passes of bench.py, kernel breakdown of the GCN / GIN steps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
bash scripts/prof_run.sh r03 > gpurun_out/final/prof_run.log 2>&1; echo "prof_run rc=$?"
python3 scripts/summarize_prof.py gpurun_out/prof_r03 r03 gcn_norm_sum_d256_BA_n10000000_m5 > gpurun_out/final/summarize.log 2>&1; echo "summarize rc=$?"; tail -3 gpurun_out/final/summarize.log
mkdir -p gpurun_out/final/profiles && cp profiles/r03_kernel_stats.csv profiles/r03_pmc.json profiles/pmc_traffic.json gpurun_out/final/profiles/
find gpurun_out/prof_r03 -type f -name "*.csv" -size +1M -delete
STEPS=8 bash scripts/prof_model.sh gcn > gpurun_out/final/prof_gcn.log 2>&1; echo "gcn rc=$?"
STEPS=8 bash scripts/prof_model.sh gin > gpurun_out/final/prof_gin.log 2>&1; echo "gin rc=$?"
head -6 gpurun_out/model_gcn/steady.csv | cut -c1-200
