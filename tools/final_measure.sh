set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/final_tests.log 2>&1 || { tail -20 gpurun_out/final_tests.log; exit 1; }
tail -2 gpurun_out/final_tests.log
bash tools/profile_cmd.sh r02_d tfw bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie || exit 2
python tools/pmc_to_traffic.py gpurun_out/prof_r02_d cfg4_50M_150bp > gpurun_out/traffic_r02_d.log 2>&1 || { cat gpurun_out/traffic_r02_d.log; exit 3; }
cp profiles/probe_hbm_bytes.json gpurun_out/probe_hbm_bytes.json
timeout -k 10 600 python bench.py > gpurun_out/bench_r02_d.json 2> gpurun_out/bench_r02_d.err || { tail -5 gpurun_out/bench_r02_d.err; exit 4; }
timeout -k 10 300 python bench.py --config cfg2_1M_150bp --no-cpu-baseline > gpurun_out/bench_r02_d_cfg2.json 2>> gpurun_out/bench_r02_d.err
timeout -k 10 300 python bench.py --config cfg3_5M_150bp --no-cpu-baseline --no-pcie > gpurun_out/bench_r02_d_cfg3.json 2>> gpurun_out/bench_r02_d.err
timeout -k 10 300 python tools/cli_e2e.py > gpurun_out/cli_e2e_r02_d.log 2>&1
tail -3 gpurun_out/cli_e2e_r02_d.log
cut -c1-600 gpurun_out/bench_r02_d.json
