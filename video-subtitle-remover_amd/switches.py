"""Switches of the dead-work eliminations that were built and replayed on the CPU in round 4 and first ran on a GPU in round 5
(profiles/r05_first_call_*.log: bit-equality tests green, A/Bs on one box).  One place decides their defaults; an environment variable
of the same name ("0" / "1") overrides.

    VSR_DECODE_COLS     STTN: the decoder / last block on the mask's columns as well as its rows (vsr_sttn_auto_chunk_box, _det_batch_box)
                        -- default ON since round 5: 216.3 -> 218.6 fps on the headline (582.2 -> 571.9 GFLOP per frame), same bits;
                        neutral on config 3 (115.3 vs 118.3 fps file to file, inside the run-to-run spread)
    VSR_PP_DECODE_BOX   ProPainter: soft composition, decoder, last transformer block on the box the plugin blends in (vsr_pp_forward_box)
                        -- default ON since round 5: 16.19 -> 17.16 fps on config 4 file to file
    VSR_PP_ENC_CACHE    ProPainter: the generator's encoder once per frame instead of once per window (vsr_pp_encode / vsr_pp_forward_cached)
                        -- default ON since round 5: 16.19 -> 16.95 fps alone, 18.06 fps with the box
    VSR_QKV0_SHARED     STTN: the first transformer block's q/k/v once per frame of a chunk instead of once per window (csrc/sttn_plan.cpp;
                        read by the library itself, once per process: _lib.py exports the default into the environment before it loads)
                        -- default ON since round 5: 216.3 -> 217.6 fps on the headline, same bits
"""
import os

DEFAULTS = {"VSR_DECODE_COLS": "1", "VSR_PP_DECODE_BOX": "1", "VSR_PP_ENC_CACHE": "1", "VSR_QKV0_SHARED": "1"}


def export_defaults():
    """the library reads its own switches from the environment (once per process): hand it the defaults that were not overridden"""
    for k, v in DEFAULTS.items():
        os.environ.setdefault(k, v)


LIBRARY_SIDE = ("VSR_DECODE_COLS", "VSR_QKV0_SHARED")      # switches libvsr_hip.so reads itself, once per process


def on(name):
    """state of a switch.  For the switches the library reads itself the answer is the LIBRARY's (vsr_switch_state: frozen at its first
    use), so that the host side can never price or promise something the engine ignores when the environment changes mid-process."""
    if name in LIBRARY_SIDE:
        from . import _lib

        v = _lib.lib.vsr_switch_state(name.encode())
        if v >= 0:
            return v == 1
    return os.environ.get(name, DEFAULTS[name]) == "1"
