python __graft_entry__.py --smoke 2>&1 | tail -2
timeout 2400 python -m pytest tests -x -q -m gpu --durations=40 2>&1 | tail -60 > gpurun_out/r06_durations.log; cat gpurun_out/r06_durations.log
