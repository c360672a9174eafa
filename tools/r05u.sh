set -u
OUT=gpurun_out/r05u; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python tools/fp64_truth.py --attribute 1 > $OUT/fp64_attribution.jsonl 2> $OUT/fp64_attribution.err; echo "attr rc=$?"; tail -1 $OUT/fp64_attribution.jsonl | cut -c1-2500
bash tools/crop_pmc.sh 256 r05u > $OUT/crop_pmc.txt 2>&1; echo "crop pmc rc=$?"; tail -4 $OUT/crop_pmc.txt
python3 profiles/summarize_crop_pmc.py gpurun_out/r05u 256 > $OUT/crop_counters_p2.json; echo "summ rc=$?"
find $OUT -name "*counter_collection.csv" -delete
bash tools/gpu_round_c5.sh r05u_c5 > $OUT/c5.txt 2>&1; echo "c5 rc=$?"; tail -5 $OUT/c5.txt
