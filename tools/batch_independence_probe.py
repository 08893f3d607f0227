import os, sys, torch
sys.path.insert(0, '.')
import voicesplit_amd as V
from oracle import reference_forward as R
dims_d = R.default_dims()
sd = R.spread_logits(R.build_state_dict(dims_d, 0), 8.0)
m = V.VoiceSplit(V.default_config()).eval(); m.load_state_dict(sd); m = m.cuda()
x, dvec = R.synthetic_inputs(8, 301, dims_d, 0)
for amp in (1.0, 3.0, 40.0):
    xb = x.clone(); xb[5] *= amp
    with torch.no_grad():
        big = m(xb.cuda(), dvec.cuda()); one = m(xb[3:4].contiguous().cuda(), dvec[3:4].contiguous().cuda())
    d = (big[3] - one[0]).abs()
    print(os.environ.get('VOICESPLIT_F16X3_CONV', 'split'), 'amp', amp, 'equal', torch.equal(big[3], one[0]), 'max diff', d.max().item(), 'n diff', int((d > 0).sum()))
