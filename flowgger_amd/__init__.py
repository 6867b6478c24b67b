"""flowgger_amd -- MI355X (gfx950) bulk log-line decoder behind flowgger's Decoder/Record interface.

Scope: the per-line ``Decoder::decode()`` hot path (RFC5424, LTSV, GELF) of awslabs/flowgger,
rebuilt as hand-written HIP kernels behind a C ABI (include/fg_hip.h).  See DESIGN.md.
"""
from .record import DecodeError, Record, SDValue, StructuredData  # noqa: F401
from .decoder import Decoder, GelfDecoder, LTSVDecoder, RFC3164Decoder, RFC5424Decoder, pack_lines  # noqa: F401
from .encoder import (Encoder, GelfEncoder, LTSVEncoder, PassthroughEncoder, Pipeline, RFC3164Encoder,  # noqa: F401
                      RFC5424Encoder, Transcoded)

__all__ = ["Decoder", "RFC5424Decoder", "RFC3164Decoder", "LTSVDecoder", "GelfDecoder", "Record", "StructuredData",
           "SDValue", "DecodeError", "pack_lines", "Encoder", "GelfEncoder", "LTSVEncoder", "RFC5424Encoder",
           "RFC3164Encoder", "PassthroughEncoder", "Pipeline", "Transcoded"]
