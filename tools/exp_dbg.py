import sys, random
sys.path[:0]=['/root/repo','/root/repo/tests']
import oracle_binding as ob
from cpecan_amd import api
from cpecan_amd.workload import make_pair
def run(probs, mtype=0, **kw):
    sm = api.stateMachine5_construct() if mtype==0 else api.stateMachine3_construct()
    acc = api.hmm_constructEmpty(1e-12, mtype)
    with api.Batch(sm, api.pairwiseAlignmentBandingParameters_construct(**kw), emit=api.EMIT_EXPECT) as b:
        for sx,sy,a in probs: b.add(sx,sy,a,True,False)
        b.upload(); print('uploaded', flush=True); b.run(); print('ran', flush=True); b.download(); print('downloaded', flush=True)
        b.expectations(acc)
    print('lik', acc.likelihood, flush=True)
sx,sy,a = make_pair(11,0,300,20)
run([(sx,sy,a)], diagonalExpansion=20)
sx,sy,a = make_pair(11,0,1300,20)
run([(sx,sy,a)], diagonalExpansion=20)
