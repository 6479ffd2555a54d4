set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_suite.log 2>&1 || { tail -40 gpurun_out/r05_gpu_suite.log; exit 1; }
tail -3 gpurun_out/r05_gpu_suite.log
python bench.py --no-extras --no-cpu-baseline 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['keyframes'])"
