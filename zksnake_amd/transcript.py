"""
Fiat-Shamir transcript with the reference's behaviour (python/zksnake/transcript.py:28-71): a running blake2b hash;
a challenge is the digest, and the digest also seeds the hasher that continues the transcript.

Byte encodings are the reference's, quirks included: an int is written big-endian on `bit_length()` BYTES (so with
leading zero bytes; 0 contributes nothing), a point as its compressed encoding, a list as the concatenation of its
items.  (hash_to_scalar / hash_to_curve of transcript.py:6-25 belong to the Bulletproofs/IPA side of the reference,
outside this backend's path.)
"""

import hashlib

from .constant import BN254_SCALAR_FIELD
from .ecc import ispointG1, ispointG2


def _encode(item) -> bytes:
    """the bytes one transcript item contributes; TypeError for anything the reference rejects"""
    if isinstance(item, bytes):
        return item
    if isinstance(item, str):
        return item.encode()
    if isinstance(item, int):
        return item.to_bytes(item.bit_length(), "big")
    if ispointG1(item) or ispointG2(item):
        return bytes(item.to_bytes())
    if isinstance(item, list) and item:
        head = item[0]
        if isinstance(head, int) or ispointG1(head) or ispointG2(head):
            return b"".join(_encode(x) for x in item)
    raise TypeError(f"Type of {type(item)} is not supported as transcript")


class FiatShamirTranscript:
    def __init__(self, label: bytes = b"", field=BN254_SCALAR_FIELD, alg="blake2b"):
        self.alg, self.label, self.field = alg, label, field
        self.state = []
        self.hasher = hashlib.new(alg, label)

    def reset(self):
        self.hasher = hashlib.new(self.alg, self.label)

    def append(self, data):
        self.hasher.update(_encode(data))

    def get_challenge(self) -> bytes:
        out = self.hasher.digest()
        self.hasher = hashlib.new(self.alg, out)
        return out

    def get_challenge_scalar(self) -> int:
        return int.from_bytes(self.get_challenge(), "big") % self.field
