MSL_PEAC_TIMING=1 timeout 300 python bench.py --config 4 --cpu-frames 0 --steps 10 2>&1 | grep "pool:\|batch of" | tail -24
