#!/bin/bash
# Hand-over tuning (round 5): PCIe-inclusive breakdown under the upload knobs, one process per setting.
out=gpurun_out/r05
mkdir -p $out
for nt in 1 0; do for kib in 512 256 128; do for thr in 12 16; do for seg in 6 4; do
  echo "== NT=$nt MIN_KIB=$kib THREADS=$thr SEG=$seg" >> $out/upload_tune.log
  COREG_UPLOAD_NT=$nt COREG_UPLOAD_MIN_KIB=$kib COREG_UPLOAD_THREADS=$thr COREG_UPLOAD_SEGMENT_MIB=$seg python profiles/pcie_breakdown.py 2>/dev/null | grep -E "set_small f32|prepare_reference f32|sweep, host|whole call" >> $out/upload_tune.log
done; done; done; done
echo "== overlap off" >> $out/upload_tune.log
COREG_OVERLAP_UPLOAD=0 python profiles/pcie_breakdown.py 2>/dev/null | grep -E "set_small f32|prepare_reference f32|sweep, host|whole call" >> $out/upload_tune.log
