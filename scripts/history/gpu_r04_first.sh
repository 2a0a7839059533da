mkdir -p gpurun_out/r04a
O=gpurun_out/r04a
rocm-smi --showclocks > $O/clocks_before.txt 2>&1
./scripts/ubench/valu_rates > $O/ubench_valu_rates.txt 2>&1
rocm-smi --showclocks > $O/clocks_after_valu.txt 2>&1
./scripts/ubench/l1_window_rate > $O/ubench_l1_window_rate.txt 2>&1
timeout 300 ./scripts/ubench/rcp_exact > $O/ubench_rcp_exact.txt 2>&1
timeout 1000 python -m pytest tests -m gpu -x -q --durations=40 > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
timeout 400 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -5 $O/pytest.txt
cat $O/bench.json | head -c 1500
