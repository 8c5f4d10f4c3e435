"""viterbidecodercpp_amd -- MI355X (gfx950) implementation of the Viterbi update()+chainback() hot path.

Product code: csrc/ (HIP kernels + C ABI, built to libvit_hip.so) and the host-side mirror of the reference's decoder
interface in decoder.py.  codes.py / synth.py hold constants and synthetic-input generation for the measurement.
"""
from .codes import COMMON_CODES, Code, DecoderConfig, get_decoding_config, SOFT16, SOFT8, HARD8  # noqa: F401
from .decoder import (BatchDecoder, DecodePipeline, ViterbiBranchTable, ViterbiDecoder_Config, ViterbiDecoder_Core,  # noqa: F401
                      ViterbiDecoder_HIP, pack_blob)
from . import _lib, dist, synth  # noqa: F401

__all__ = ["COMMON_CODES", "Code", "DecoderConfig", "get_decoding_config", "SOFT16", "SOFT8", "HARD8", "BatchDecoder", "DecodePipeline",
           "ViterbiBranchTable", "ViterbiDecoder_Config", "ViterbiDecoder_Core", "ViterbiDecoder_HIP", "pack_blob"]
