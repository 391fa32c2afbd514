#!/bin/bash
# encoder parity tests on the shipped build, then a development build for interleaved A/B of the kernel variants,
# on uniform and on sensor-ordered clouds
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02d}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu > $O/pytest_enc.log 2>&1; echo "rc=$?" >> $O/pytest_enc.log
tail -4 $O/pytest_enc.log
NSC_DEV_BUILD=1 python neural-spectral-codec_amd/build.py > $O/devbuild.log 2>&1
shift
for order in uniform azimuth_major ring_major; do
  echo "== order $order" >> $O/ab.log
  AB_ORDER=$order timeout -k 10 600 python tools/ab_enc.py "$@" >> $O/ab.log 2>&1
done
cat $O/ab.log
