"""Share validation: N*B per-share PublicKeyShare::verify checks (tc_verify_sig_batch, pk_stride = 96) vs one
random linear combination per message (tc_verify_shares_rlc_batch), device-resident operands.
usage: python tools/rlc_probe.py [B] [N]   -> one JSON line (profiles/r02_rlc_probe.txt)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from threshold_crypto_amd.engine import Engine, pack_messages
from threshold_crypto_amd.workload import key_set, messages

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
t = 3
e = Engine(0)
sks = key_set(t)
fr = np.stack([np.frombuffer(sks.secret_key_share(i)._bytes(), dtype=np.uint8) for i in range(N)])
msgs = messages(B)
flat, off = pack_messages(msgs)
sig, st = e.sign(fr, flat, off)
commit = np.stack([np.frombuffer(c, dtype=np.uint8) for c in sks.public_keys(e).commit])
pks, _ = e.public_key_shares(commit, np.arange(N, dtype=np.uint64))
dev = torch.device("cuda", 0)
d_sig = torch.from_numpy(sig).to(dev)
d_pks = torch.from_numpy(pks).to(dev)
d_flat, d_off = torch.from_numpy(flat).to(dev), torch.from_numpy(off.view(np.int64)).to(dev)
rep_flat, rep_off = pack_messages([m for m in msgs for _ in range(N)])
d_rflat, d_roff = torch.from_numpy(rep_flat).to(dev), torch.from_numpy(rep_off.view(np.int64)).to(dev)
d_pk_rep = d_pks[None].expand(B, N, 96).reshape(B * N, 96).contiguous()
d_sig_flat = d_sig.reshape(B * N, 192).contiguous()
res = {"B_messages": B, "N_shares": N}
for planted in (0, max(1, B // 256)):
    s2 = d_sig.clone()
    if planted:
        js = torch.arange(0, B, B // planted, device=dev)[:planted]
        s2[js, 1] = d_sig[js, 2]
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ok_ps = e.verify_sig(d_pk_rep, s2.reshape(B * N, 192).contiguous(), d_rflat, d_roff); e.sync(); torch.cuda.synchronize()
        t_ps = time.perf_counter() - t0
        t0 = time.perf_counter()
        ok_rlc, nfb = e.verify_shares_rlc(d_pks, s2, d_flat, d_off); e.sync(); torch.cuda.synchronize()
        t_rlc = time.perf_counter() - t0
    assert bool((ok_ps.reshape(B, N) == ok_rlc).all().item())
    assert int((ok_rlc == 0).sum().item()) == planted and nfb == planted
    k = "all_valid" if not planted else "bad_share_in_%d_messages" % planted
    res[k] = {"per_share_ms": round(t_ps * 1e3, 2), "rlc_ms": round(t_rlc * 1e3, 2), "speedup": round(t_ps / t_rlc, 2),
              "share_verifies_per_s_per_share_path": round(B * N / t_ps, 1), "share_verifies_per_s_rlc": round(B * N / t_rlc, 1),
              "messages_in_fallback": nfb}
print(json.dumps(res), flush=True)
