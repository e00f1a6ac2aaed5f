for extra in "" "--force-collective-path --no-overlap" "--force-collective-path"; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-extras $extra 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('%-40s %.4f ms  %s' % ('$extra' or 'single graph', r['ms_per_step'], r['config']['launch_mode'][:50]))"
done
