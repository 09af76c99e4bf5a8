import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
import sionna_amd.phy as phy
k, n, m = 2816, 8448, 6
for B in (4096, 16384):
    phy.config.seed = 3
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
    no = phy.utils.ebnodb2no(4.5, m, k / n)
    u = phy.mapping.BinarySource()([B, k])
    llr = phy.mapping.Demapper("app", "qam", m)(phy.channel.AWGN()(phy.mapping.Mapper("qam", m)(enc(u)), no), no)
    for cn in ("minsum", "boxplus-phi"):
        dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, num_iter=5, hard_out=False, return_state=True)
        ref = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, num_iter=20, hard_out=False)
        def idd():
            x, st = dec(llr)
            for _ in range(3):
                x, st = dec(llr, msg_v2c=st)
            return x
        x = idd(); torch.cuda.synchronize()
        y = ref(llr); torch.cuda.synchronize()
        same = bool(torch.equal(x.as_subclass(torch.Tensor), y.as_subclass(torch.Tensor)))
        t0 = time.perf_counter()
        for _ in range(3): idd()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 3
        t0 = time.perf_counter()
        for _ in range(3): ref(llr)
        torch.cuda.synchronize()
        tr = (time.perf_counter() - t0) / 3
        print(f"B={B} {cn}: 4 calls x 5 iterations with state {t*1e3:.2f} ms = {B/t/1e3:.1f} k decodes/s; one call x 20 iterations on chip {tr*1e3:.2f} ms = {B/tr/1e3:.1f} k; same soft outputs: {same}", flush=True)
