#!/bin/bash
# GPU box: the residual of the captured steps (every labelled launch dropped) -> OUT.json   usage: tools/residual.sh OUT.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUTJ=$1; case $OUTJ in /*) ;; *) OUTJ=$R/$OUTJ;; esac
rm -f $OUTJ
for prec in fp32 bf16; do for kind in G D; do
  D=/tmp/resid_$prec$kind; rm -rf $D
  rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/tools/trace_residual.py $prec $kind 20 2>&1 | grep residual
  python3 $R/tools/residual_summary.py $(ls $D/*/*kernel_trace.csv | head -1) $OUTJ ${prec}_$kind
done; done
