import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import sionna_amd.phy as phy
from oracle.ldpc5g import LDPC5GCode
import oracle.ldpc_bp as obp
for (k,n,bg) in [(1024,2048,"bg1"),(64,128,None)]:
    code = LDPC5GCode(k,n,None,bg)
    enc = phy.fec.ldpc.LDPC5GEncoder(k,n,bg=bg)
    rng=np.random.default_rng(3)
    u=rng.integers(0,2,(3,k)).astype(np.float32)
    c=code.encode(u)
    llr=((2*c-1)*2+rng.standard_normal(c.shape)*1.2).astype(np.float32)
    z=code.z
    odec=obp.LDPC5GDecoder(code,cn_update="boxplus-phi",num_iter=1)
    pcm=odec.pcm if hasattr(odec,'pcm') else odec._pcm
    pcm=pcm.tocsr()
    for it in (1,):
        # emulate the n-output -> VN map by decoding with return_infobits False and mapping back is messy; use the generic API on the recovered llr
        dec=phy.fec.ldpc.LDPC5GDecoder(enc,cn_update="boxplus-phi",hard_out=False,return_infobits=False,num_iter=it)
        a=dec(llr).cpu().numpy(); dec._onchip_ok=False; b=dec(llr).cpu().numpy()
        idx=np.nonzero(a[0]!=b[0])[0]
        full=idx+2*z   # no interleaver, position in codeword (k>=? fillers shift parity part)
        full=np.where(full<k, full, full+(code.k_ldpc-k))
        cols=full//z; lanes=full%z
        print(k,n,"z",z,"k_ldpc",code.k_ldpc,"wrong VN count",len(idx))
        for cc in sorted(set(cols.tolist())):
            l=lanes[cols==cc]; print("   col",cc,"lanes",l.min(),"..",l.max(),"count",len(l))
        # rows touching these columns
        rows=set()
        for v in full[:5]:
            rows |= set(pcm[:, v].nonzero()[0]//z)
        print("   candidate base rows of first wrong VNs", sorted(rows))
        for r in sorted(rows)[:12]:
            vs=pcm[r*z].nonzero()[1]
            print("     row",r,"deg",len(vs),"cols",(vs//z).tolist(),"shifts",(vs%z).tolist())
