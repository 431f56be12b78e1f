"""One leg of the bench line, alone, for rocprofv3 (tools/capture_legs.sh): the workload is set up first, then the leg runs `reps`
times as the LAST launches of its kernels in the process -- tools/rocpd_summary.py --last <reps> then averages exactly those
launches (duration, FETCH_SIZE, WRITE_SIZE, SQ_*), so a kernel's per-launch figures describe ONE kind of launch.
    python tools/profile_legs.py <leg> [reps]
legs: combine | verify_g2 | hash_g2 | g2_sign | ciphertext_verify | threshold_decrypt | wire | general_path
BASELINE configs 2-4 at their stated batch (t = 3, N = 10, 65 536 jobs), operands resident in HBM, input checks off (the
operands are outputs of the library's own kernels), as in bench.py."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload, ThresholdEncWorkload

leg = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
t, N, B = 3, 10, int(os.environ.get("PROBE_B", "65536"))
e = Engine(0); e.set_timing(True); e.set_input_checks(False)
dev = torch.device("cuda:0")
to = lambda a: torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).to(dev)
ms = []
if leg in ("threshold_decrypt", "ciphertext_verify"):
    we = ThresholdEncWorkload(e, t, N, B)
    du, dv, dw, doff, didx, dsh = to(we.u), to(we.v), to(we.w), to(we.off), to(we.idx), to(we.shares)
    for _ in range(reps):
        if leg == "ciphertext_verify":
            ok = e.ciphertext_verify(du, dv, doff, dw)
        else:
            out, st = e.decrypt(t, didx, dsh, dv, doff)
        ms.append(e.last_kernel_ms())
    assert (int(ok.to(torch.int32).sum().item()) == B) if leg == "ciphertext_verify" else (int(st.to(torch.int32).sum().item()) == 0)
else:
    offset = (1 << 20) if leg == "general_path" else 0
    wl = ThresholdSigWorkload(e, t, N, B, index_offset=offset) if offset else ThresholdSigWorkload(e, t, N, B)
    d_idx, d_sh, d_hash = to(wl.idx), to(wl.shares), to(wl.hashes)
    pk = to(np.ascontiguousarray(wl.master_pk))
    sig, st = e.combine_g2(t, d_idx, d_sh)
    e.sync()
    if leg == "wire":
        comp, _ = e.g2_compress(d_sh.reshape(B * (t + 1), 192))
        d_wire = comp.reshape(B, t + 1, 96).contiguous()
    if leg == "hash_g2":
        d_msgs, d_off = to(wl.msg_flat), to(wl.msg_off)
    if leg == "g2_sign":
        d_sk = to(np.stack([np.frombuffer(wl.shares_sk[i]._bytes(), dtype=np.uint8) for i in range(t + 1)]))
    e.sync()
    for _ in range(reps):
        if leg in ("combine", "general_path"):
            sig, st = e.combine_g2(t, d_idx, d_sh)
        elif leg == "verify_g2":
            ok = e.verify_g2(pk, sig, d_hash)
        elif leg == "hash_g2":
            h = e.hash_g2(d_msgs, d_off)
        elif leg == "g2_sign":
            sh, st2 = e.g2_mul(d_sk, d_hash)
        elif leg == "wire":
            wsig, wst = e.combine_signatures_wire(t, d_idx, d_wire)
        else:
            raise SystemExit("unknown leg " + leg)
        ms.append(e.last_kernel_ms())
    if leg == "verify_g2":
        assert int(ok.to(torch.int32).sum().item()) == B
e.sync()
print(json.dumps({"leg": leg, "jobs": B, "reps": reps, "kernel_ms": [round(x, 3) for x in ms]}), flush=True)
