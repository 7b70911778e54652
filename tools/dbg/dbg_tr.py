import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'video-steganography-pcamv_amd')
import helpers, pcamv_amd as pc
import test_gpu_parity as T
g=helpers.load('qcif_hex_subme5')
enc=pc.Encoder(T._fixture_params(pc,g))
t=1
enc.set_ref(g[f"f{t}_ref_y"], g[f"f{t}_ref_u"], g[f"f{t}_ref_v"]); enc.upload_fenc(g[f"f{t}_fenc_y"], g[f"f{t}_fenc_u"], g[f"f{t}_fenc_v"])
mbs,rec=enc.analyse_pframe(int(g["qp"]),embed=1)
for k,nm in enumerate("yuv"):
    a=rec[k].astype(int); b=g[f"f{t}_rec_{nm}"].astype(int)
    d=np.argwhere(a!=b)
    print(nm,len(d))
    if len(d):
        y,x=d[0]; y0=y&~3; x0=x&~3
        print('first',y,x,'block',y0,x0); print(a[y0:y0+4,x0:x0+4]-b[y0:y0+4,x0:x0+4])
        # mb / block stats
        blocks={(yy>>2,xx>>2) for yy,xx in d}
        print('blocks differing',len(blocks), sorted(blocks)[:10])
