mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_surfel_gpu.py tests/test_clutter_gpu.py -x -q -m gpu > gpurun_out/t_sf.log 2>&1; tail -3 gpurun_out/t_sf.log
for d in 1 0; do
  MSL_SF_DEAL=$d timeout 300 python bench.py --cpu-frames 0 --no-breakdown --no-parity-gate --steps 8 > gpurun_out/ab_deal$d.json 2> gpurun_out/ab_deal$d.err
  python -c "
import json; d=json.load(open('gpurun_out/ab_deal$d.json')); r=d['roofline']; print('deal=$d', d['value'], r['frac'], r['avg_launch_us'], r['avg_launch_us_event_pair_raw'], r['event_pair_empty_kernel_us'])"
done
for d in 1 0; do MSL_SF_DEAL=$d MSL_SF_DEFER=0 timeout 200 python tools/fuse_iso.py 3 2>/dev/null | tail -1; done
