#!/bin/bash
# GPU box: the bench line under different settings of ONE environment variable, interleaved over rounds; prints value / value_sync and the
# per-step time of the op families matching a pattern.   tools/micro/env_ab.sh VAR "v1 v2 ..." <rounds> "<family regex>"   (extra flags: $YMK_AB_FLAGS)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
VAR=$1; VALS=$2; ROUNDS=$3; PAT=$4
rm -f /tmp/env_ab_lines.txt
for r in $(seq 1 $ROUNDS); do
  for v in $VALS; do
    env $VAR=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline $YMK_AB_FLAGS 2>/dev/null | python -c "
import json,sys,re
r=json.loads(sys.stdin.read())
f={x['kernel']:(x['ms_per_step'],x['launches_per_step']) for x in (r.get('families') or []) if re.search(r'$PAT', x['kernel'])}
print('round $r $VAR=$v value', r['value'], 'sync', r['value_sync'], 'ms', r['ms_per_step'], 'calls', r['op_calls_per_step'], f, 'sum %.4f' % sum(v[0] for v in f.values()))" | tee -a /tmp/env_ab_lines.txt
  done
done
python - <<'PY'
import re, collections
d = collections.defaultdict(lambda: ([], []))
for l in open('/tmp/env_ab_lines.txt'):
    m = re.search(r'round \d+ (\S+) value ([0-9.]+) sync ([0-9.]+)', l)
    if m:
        d[m.group(1)][0].append(float(m.group(2))); d[m.group(1)][1].append(float(m.group(3)))
med = lambda v: sorted(v)[len(v) // 2]
for k, (a, b) in d.items():
    print(f'median {k}: value {med(a):.0f} (min {min(a):.0f} max {max(a):.0f})  value_sync {med(b):.0f} (min {min(b):.0f} max {max(b):.0f})  n={len(a)}')
PY
