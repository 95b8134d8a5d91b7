cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06e
(time timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r06e/pytest_gpu_all.txt 2>&1
python bench.py --config 4s > gpurun_out/r06e/c4s_dev.json 2> gpurun_out/r06e/c4s_dev.err
python bench.py --config 4s --sessions S02 > gpurun_out/r06e/c4s_s02.json 2> gpurun_out/r06e/c4s_s02.err
python bench.py --config 4s --gpus 2 > gpurun_out/r06e/c4s_dev_g2.json 2> gpurun_out/r06e/c4s_dev_g2.err
cat gpurun_out/r06e/pytest_gpu_all.txt
for f in gpurun_out/r06e/c4s_*.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=d['config4_standin']
print(sys.argv[1], 'value %.1f'%d['value'], 'gpu_wait_share', b['gpu_wait_share_of_wall'], b.get('utterances_per_session'), 'wall', b['wall_s'])
PY
done
tail -3 gpurun_out/r06e/*.err
