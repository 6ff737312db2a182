"""
Fiat-Shamir transcript of the reference (python/zksnake/transcript.py:28-71): a running blake2b hash;
a challenge is the digest, which also seeds the next hasher.  Byte encodings are the reference's, quirks
included: an int is written big-endian on `bit_length()` BYTES (so with leading zero bytes), a point as
its compressed encoding.  (hash_to_scalar / hash_to_curve of transcript.py:6-25 belong to the
Bulletproofs/IPA side of the reference, outside this backend's path.)
"""

import hashlib

from .constant import BN254_SCALAR_FIELD
from .ecc import ispointG1, ispointG2


class FiatShamirTranscript:
    def __init__(self, label: bytes = b"", field=BN254_SCALAR_FIELD, alg="blake2b"):
        self.alg = alg
        self.label = label
        self.hasher = hashlib.new(alg, label)
        self.state = []
        self.field = field

    def reset(self):
        self.hasher = hashlib.new(self.alg, self.label)

    @staticmethod
    def _int_bytes(d):
        return int.to_bytes(d, d.bit_length(), "big")

    def append(self, data):
        if isinstance(data, bytes):
            self.hasher.update(data)
        elif isinstance(data, str):
            self.hasher.update(data.encode())
        elif isinstance(data, int):
            self.hasher.update(self._int_bytes(data))
        elif ispointG1(data) or ispointG2(data):
            self.hasher.update(bytes(data.to_bytes()))
        elif data and isinstance(data, list) and isinstance(data[0], int):
            for d in data:
                self.hasher.update(self._int_bytes(d))
        elif data and isinstance(data, list) and (ispointG1(data[0]) or ispointG2(data[0])):
            for d in data:
                self.hasher.update(bytes(d.to_bytes()))
        else:
            raise TypeError(f"Type of {type(data)} is not supported as transcript")

    def get_challenge(self) -> bytes:
        digest = self.hasher.digest()
        self.hasher = hashlib.new(self.alg, digest)
        return digest

    def get_challenge_scalar(self) -> int:
        return int.from_bytes(self.get_challenge(), "big") % self.field
