#!/bin/bash
# usage: r02_ab_only.sh TAG cfg...
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02e}
mkdir -p $O
cd $R
NSC_DEV_BUILD=1 python neural-spectral-codec_amd/build.py > $O/devbuild.log 2>&1
shift
timeout -k 10 900 python tools/ab_enc.py "$@" >> $O/ab.log 2>&1
cat $O/ab.log
