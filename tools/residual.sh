#!/bin/bash
# GPU box: the residual of the captured steps (every labelled launch dropped) -> OUT.json   usage: tools/residual.sh OUT.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUTJ=$1; case $OUTJ in /*) ;; *) OUTJ=$R/$OUTJ;; esac
rm -f $OUTJ /tmp/resid_wall.log
for prec in fp32 bf16; do for kind in G D; do
  D=/tmp/resid_$prec$kind; rm -rf $D
  rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/tools/trace_residual.py $prec $kind 20 2>&1 | grep residual | tee -a /tmp/resid_wall.log
  python3 $R/tools/residual_summary.py $(ls $D/*/*kernel_trace.csv | head -1) $OUTJ ${prec}_$kind
done; done
# host-clock figures of the same replays (ts.step() wall / inputs unchanged / bare graph replay) into the same file
python3 - "$OUTJ" <<'PY'
import json, re, sys
r = json.load(open(sys.argv[1]))
for m in re.finditer(r'residual (\w+) ([GD])-step(, [\w ]+)?: ([\d.]+) ms per replay', open('/tmp/resid_wall.log').read()):
  what = (m.group(3) or ', ts.step() wall').strip(', ')
  r.setdefault('%s_%s' % (m.group(1), m.group(2)), {}).setdefault('wall_ms_per_replay', {})[what] = float(m.group(4))
json.dump(r, open(sys.argv[1], 'w'), indent=1)
PY
