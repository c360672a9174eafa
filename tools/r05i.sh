set -u
OUT=gpurun_out/r05i; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/determinism_probe.py 300 > $OUT/det_default.txt 2>&1; cat $OUT/det_default.txt | cut -c1-300
MRCNN_W4_VMCNT0=1 timeout -k 10 400 python maskrcnn_amd/build.py > $OUT/build.log 2>&1; echo "rebuild rc=$?"
timeout -k 10 300 python tools/determinism_probe.py 300 > $OUT/det_vmcnt0.txt 2>&1; cat $OUT/det_vmcnt0.txt | cut -c1-300
