"""32x32 generator / critic with the reference's surface (src/networks/conv32.py:9-82): first/last kernel 2."""
from .convnets import _Decoder, _Encoder


class Decoder(_Decoder):
    K0 = 2


class Encoder(_Encoder):
    K0 = 2
