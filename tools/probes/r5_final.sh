#!/bin/bash
# round 5, last checks: smoke(), the RCCL path with one rank, FastGA -G with virtual ranks at 3 Gbp, then the whole GPU suite
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5z; mkdir -p $o
export TMPDIR=/tmp
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $o/smoke.log 2>&1; tail -1 $o/smoke.log
timeout 600 python bench.py --gpus 1 --force-sharded --steps 5 --warmup 2 --no-human-scale > $o/bench_sharded.json 2> $o/bench_sharded.err; python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r5z/bench_sharded.json').read().strip().splitlines()[-1])
    print('sharded(1 rank): ms/step %.2f value %.3f parity %s' % (d['ms_per_step'], d['value'], json.dumps(d.get('parity'))[:200]))
except Exception as e:
    print('sharded failed', e); print(open('gpurun_out/r5z/bench_sharded.err').read()[-600:])
PY
( timeout 900 python tools/multi_3g_check.py 2>&1 | tail -6 ) > $o/multi3g.log 2>&1; tail -4 $o/multi3g.log | cut -c1-300
( time timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 ) > $o/suite.txt 2>&1; tail -6 $o/suite.txt
