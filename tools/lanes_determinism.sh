# headline mode (exact + masks_only) through WindowPipeline with 1 and 2 lanes: features / masks hashes of 32 windows x 4 reps
cd $GRAFT_REPO_ROOT
for L in 2 1; do
  VIDSEG_DEBUG_HASH=1 timeout 900 python tools/determinism_check.py --precision exact --overlap --lanes $L --masks-only --windows 0-31 --reps 4 > /tmp/det_$L.out 2> /tmp/det_$L.err
  grep "^HASH" /tmp/det_$L.err | awk '{print $2, $4, $6}' | sort | uniq -c | awk '{c[$2]++; n[$2]+=$1} END {bad=0; for (w in c) if (c[w] > 1) {bad++; print "  window", w, "has", c[w], "distinct (features, masks) hash pairs over", n[w], "runs"}; print "lanes='$L': windows with more than one hash:", bad, "of", length(c)}'
done
# the 16-bit mode (bench.py's fast_mode, two lanes): its attention kernels carry packed fp32 VALU ops too (profiles/KERNEL_NOTES.md, round 6)
VIDSEG_DEBUG_HASH=1 timeout 600 python tools/determinism_check.py --precision fp16 --overlap --lanes 2 --windows 0-31 --reps 4 > /tmp/det_f.out 2> /tmp/det_f.err
grep "^HASH" /tmp/det_f.err | awk '{print $2, $4, $6}' | sort | uniq -c | awk '{c[$2]++; n[$2]+=$1} END {bad=0; for (w in c) if (c[w] > 1) {bad++; print "  window", w, "has", c[w], "distinct (features, masks) hash pairs over", n[w], "runs"}; print "16-bit mode, lanes=2: windows with more than one hash:", bad, "of", length(c)}'
