# per-launch times of the detector plan under the tuning switches of ocr_det_nhwc._opts
# (the sweep that chose ocr_det_nhwc.thin_variant: VSR_DET_THIN_K / _THIN_VARIANT / _N96_TILE existed for this sweep only and were folded into the rule afterwards;
#  its output is profiles/r06c_det_plan_variants.log)
run() { echo "=== $*"; env "$@" DET_AB_CASES=ppocr_det_graph.json:8 DET_AB_ONLY=plan DET_AB_STEPS=8 python scripts/r06/det_nhwc_ab.py 2>&1 | grep -v amdgpu.ids | head -${HEADN:-45}; }
run VSR_DET_N32_TILE=1 VSR_DET_THIN_K=256 VSR_DET_THIN_VARIANT=1 VSR_DET_N96_TILE=1
run VSR_DET_N32_TILE=1 VSR_DET_THIN_K=576 VSR_DET_THIN_VARIANT=1
run VSR_DET_N32_TILE=1 VSR_DET_THIN_K=1152 VSR_DET_THIN_VARIANT=1
run VSR_DET_N32_TILE=1 VSR_DET_THIN_K=2400 VSR_DET_THIN_VARIANT=1
