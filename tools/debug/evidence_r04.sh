set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ev
timeout -k 10 400 python bench.py > gpurun_out/ev/bench_full.log 2>&1
tail -1 gpurun_out/ev/bench_full.log > gpurun_out/ev/r04_e_bench_line.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ev/prof_step -- python3 bench.py --no-extras --no-cpu-baseline > gpurun_out/ev/prof_step.log 2>&1
cp $(find gpurun_out/ev/prof_step -name "*kernel_stats.csv" | head -1) gpurun_out/ev/r04_f_step_kernel_stats.csv
grep -h "^{\"metric\"" gpurun_out/ev/prof_step.log > gpurun_out/ev/r04_f_bench_line.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/ev/pmc_f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/ev/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/ev/pmc_w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/ev/pmc_w.log 2>&1
python3 tools/pmc_summary.py $(find gpurun_out/ev/pmc_f -name "*counter_collection.csv" | head -1) $(find gpurun_out/ev/pmc_w -name "*counter_collection.csv" | head -1) "gemm_nt_bf16_8phase_kernel|gemm_nt_bf16_tall_kernel" gpurun_out/ev/r04_pmc_gemm.json > gpurun_out/ev/pmc_sum.log 2>&1
timeout -k 10 300 python tools/gemm_bench.py 5536 cold > gpurun_out/ev/r04_gemm_cold_table_final.txt 2>&1
# drop the bulky traces, keep summaries
rm -rf gpurun_out/ev/prof_step gpurun_out/ev/pmc_f gpurun_out/ev/pmc_w gpurun_out/ev/prof_dec
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ev/prof_unf -- python3 bench.py --mode unfrozen --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/ev/prof_unf.log 2>&1
cp $(find gpurun_out/ev/prof_unf -name "*kernel_stats.csv" | head -1) gpurun_out/ev/r04_g_unfrozen_kernel_stats.csv
grep -h "^{\"metric\"" gpurun_out/ev/prof_unf.log > gpurun_out/ev/r04_g_unfrozen_bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ev/prof_dec -- python3 tools/bench_decode.py > gpurun_out/ev/prof_dec.log 2>&1
cp $(find gpurun_out/ev/prof_dec -name "*kernel_stats.csv" | head -1) gpurun_out/ev/r04_h_decode_kernel_stats.csv
rm -rf gpurun_out/ev/prof_unf gpurun_out/ev/prof_dec
