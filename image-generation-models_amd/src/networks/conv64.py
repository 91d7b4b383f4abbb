"""64x64 generator / critic with the reference's surface (src/networks/conv64.py:8-84): first/last kernel 4."""
from .convnets import _Decoder, _Encoder


class Decoder(_Decoder):
    K0 = 4


class Encoder(_Encoder):
    K0 = 4
