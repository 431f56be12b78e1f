"""Loader of tests/golden/ref_v0.4.0/vectors.hex (records printed by tools/ref_fixtures against the real
crate).  Absent in this repository: no Rust toolchain in the build image.  source() is what
reports print next to parity claims."""
import os

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_v0.4.0", "vectors.hex")


def present():
    return os.path.exists(PATH)


def source():
    return "reference (threshold_crypto 0.4.0, tests/golden/ref_v0.4.0/vectors.hex)" if present() else "self-oracle"


def load():
    """-> {kind: [ {field: bytes-or-str} ]}"""
    out = {}
    with open(PATH) as f:
        for line in f:
            parts = line.split()
            if not parts or parts[0].startswith("#"):
                continue
            rec = {}
            for kv in parts[1:]:
                k, _, v = kv.partition("=")
                rec[k] = bytes.fromhex(v)
            out.setdefault(parts[0], []).append(rec)
    return out


def split_ciphertext(blob):
    """bincode of Ciphertext(G1, Vec<u8>, G2): 48 B || u64 LE len || v || 96 B  (src/serde_impl.rs:174-185)"""
    n = int.from_bytes(blob[48:56], "little")
    if len(blob) != 48 + 8 + n + 96:
        raise ValueError("unexpected bincode layout")
    return blob[:48], blob[56:56 + n], blob[56 + n:]


def fr_le(sk_be):
    return bytes(reversed(sk_be))
