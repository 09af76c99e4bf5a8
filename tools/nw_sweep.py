import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import sionna_amd.phy as phy
from sionna_amd import _ffi
_ffi.device()
k, n, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
phy.config.seed = 1
enc = phy.fec.ldpc.LDPC5GEncoder(k, n)
no = phy.utils.ebnodb2no(3.0, 2, k / n)
u = phy.mapping.BinarySource()([B, k])
llr = phy.mapping.Demapper("app", "qam", 2)(phy.channel.AWGN()(phy.mapping.Mapper("qam", 2)(enc(u)), no), no)
dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", num_iter=20)
out = dec(llr); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): dec(llr)
torch.cuda.synchronize()
print(os.environ.get("SAMD_ONCHIP_BP_WAVES", "auto"), round((time.perf_counter() - t0) / 5 * 1e3, 3), "ms", float(out.sum()))
