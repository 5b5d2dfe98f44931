import sys; sys.path[:0]=['/root/repo','/root/repo/open-hummingbird-eval_amd','/root/repo/tests']
import numpy as np, torch, oracle, golden_inputs as gi
from hbird_mi.nn.search_hip import HipFlatIndex
for (M,D,nq,k,metric) in [(20000,384,520,30,0),(9000,768,300,10,0),(20000,384,520,30,1),(5000,256,100,60,0)]:
    bank=gi.unit_bank(M,D,seed=M+1); q=gi.vit_like_queries(nq,D,seed=nq+1)
    ix=HipFlatIndex(D,metric,0); ix.add(bank); ix.set_fp16(True); ix.set_variant(2)
    for G,P in ((0,0),(6,2)):
        ix.set_tuning(G,P)
        idx,dist=ix.search(torch.from_numpy(q).cuda(),k)
        ri,rd=oracle.knn_chain_f32(q,bank,k,'dot_product' if metric==0 else 'l2')
        print(M,D,nq,k,metric,G,'idx ok',np.array_equal(idx.cpu().numpy(),ri),'dist ok',np.array_equal(dist.cpu().numpy().view(np.uint32),rd.view(np.uint32)),'fb',ix.last_fp16_fallbacks())
