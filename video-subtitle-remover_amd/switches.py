"""Switches of code paths that are built and replayed against the full plans on the CPU but have not run on a GPU yet.

One place decides their defaults: an environment variable of the same name ("0" / "1") overrides.  Round 5's first GPU call
(scripts/r05/first_call.sh) runs the bit-equality tests and the A/Bs with the switches on; what is green and faster becomes "1" here.

    VSR_DECODE_COLS     STTN: the decoder / last block on the mask's columns as well as its rows (vsr_sttn_auto_chunk_box, _det_batch_box)
    VSR_PP_DECODE_BOX   ProPainter: soft composition, decoder, last transformer block on the box the plugin blends in (vsr_pp_forward_box)
    VSR_PP_ENC_CACHE    ProPainter: the generator's encoder once per frame instead of once per window (vsr_pp_encode / vsr_pp_forward_cached)
    VSR_QKV0_SHARED     STTN: the first transformer block's q/k/v once per frame of a chunk instead of once per window (csrc/sttn_plan.cpp;
                        read by the library itself, once per process: _lib.py exports the default into the environment before it loads)
"""
import os

DEFAULTS = {"VSR_DECODE_COLS": "0", "VSR_PP_DECODE_BOX": "0", "VSR_PP_ENC_CACHE": "0", "VSR_QKV0_SHARED": "0"}


def export_defaults():
    """the library reads its own switches from the environment (once per process): hand it the defaults that were not overridden"""
    for k, v in DEFAULTS.items():
        os.environ.setdefault(k, v)


def on(name):
    return os.environ.get(name, DEFAULTS[name]) == "1"
