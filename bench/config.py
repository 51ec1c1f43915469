"""bench/config.py — constants of bench.py: image shape, vendor peaks, the algorithmic-byte accounting of SURVEY.md section 8(d), and the map from
profiling slots (csrc/prof.hip) to kernel symbols.  Split out of bench.py in round 5 (no behaviour change)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

H, W = 376, 1241
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_TLANEOPS = 39.3      # spec-derived: 256 CUs x 4 SIMDs x 16 lanes per cycle x 2.4 GHz (packed-16 / VOP3 integer classes: 4 cycles per wave64)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 dense
# The MEASURED peaks of the same machine class live in profiles/r<NN>_peaks.json (tools/peaks.hip via tools/peaks.py): HBM copy rate,
# issue rate of the instruction classes the FAST kernel is made of, bf16 MFMA rate.  The roofline objects carry both: `peak` is the
# vendor figure the contract names (HBM) or the measured ceiling (VALU: no vendor figure exists), the other one sits beside it.
PYR_PX = 1444097               # sum of the 8 level areas
# algorithmic bytes per IMAGE of each ORB stage (SURVEY.md §8(d) accounting)
ALGO_BYTES = {
    "resize": 1407767 + 977481,            # read levels 0-6, write levels 1-7
    "fast": PYR_PX + 4 * 20000,            # read every level once + candidate list
    "blur7": 2 * PYR_PX,                   # read + write every level
    "describe": 2000 * (749 + 512 + 60),   # IC patch + BRIEF samples + outputs
    "octree": 2 * 4 * 56000,               # candidates in, selected out (latency bound in practice)
}
# profiling slot (csrc/prof.hip) -> kernel symbol prefix as rocprofv3 prints it
SYMBOL = {"resize": "k_resize_strip", "fast": "k_fast_strip", "octree": "k_octree", "blur7": "k_blur7", "describe": "k_describe2",
          "hamming_match": "k_hamming_fp4", "triangulate": "k_triangulate", "lcd_preproc": "k_lcd_input_fused",
          "calc_conv1": "k_conv1_f16x3_pool_lrn", "calc_conv2": "k_conv2_f16x3", "calc_pool2": "k_pool_lrn128_2x2", "calc_conv3": "k_conv3_norm", "lcddb_scan": "k_db_scan_bf16x6",
          "ba_build": "k_ba_build", "screen": "k_screen"}
