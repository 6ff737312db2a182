"""Proving-key file round trip: ProvingKey.to_bytes / from_bytes at a given constraint count, then a proof from the
re-read key.  usage: python tools/key_io_bench.py [log_n] [curve]"""
import sys
import time

sys.path.insert(0, ".")
from zksnake_amd import workloads as W  # noqa: E402
from zksnake_amd.arithmetization import R1CS  # noqa: E402
from zksnake_amd.groth16 import Groth16, ProvingKey, VerifyingKey  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
curve = sys.argv[2] if len(sys.argv) > 2 else "BN254"
n = 1 << log_n
A, B, C, w, n_col = W.chain_circuit(n, W.scalar_field(curve))
r1cs = R1CS.from_triplets(A, B, C, n, n_col, 2, curve)
pub, priv = w[:2], w[2:]
g = Groth16(r1cs, curve)
t = time.perf_counter(); g.setup(); t_setup = time.perf_counter() - t
t = time.perf_counter(); kb = g.proving_key.to_bytes(); t_ser = time.perf_counter() - t
t = time.perf_counter(); pk = ProvingKey.from_bytes(kb, curve); t_de = time.perf_counter() - t
vk = VerifyingKey.from_bytes(g.verifying_key.to_bytes(), curve)
g2 = Groth16(r1cs, curve)
g2.proving_key, g2.verifying_key = pk, vk
t = time.perf_counter(); g2.prepare_prover(); t_prep = time.perf_counter() - t
t = time.perf_counter(); proof = g2.prove(pub, priv); t_first = time.perf_counter() - t
t = time.perf_counter(); proof = g2.prove(pub, priv); t_second = time.perf_counter() - t
print(f"n=2^{log_n} {curve}: setup {t_setup:.3f}s  key {len(kb)/1e6:.1f} MB  to_bytes {t_ser:.3f}s  from_bytes {t_de:.3f}s  "
      f"prepare_prover {t_prep:.3f}s  first prove {t_first*1e3:.1f} ms  second {t_second*1e3:.1f} ms  verify {g2.verify(proof, pub)}")
