"""Development aid (GPU box): the training curves of tests/test_gpu_recipe.py for a learning rate -- train loss per step, validation loss through
the evaluation step and through the training path.   python scripts/recipe_probe.py N_TRAIN LR MOMENTUM EPOCHS"""
import sys, pathlib, tempfile, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch
from torchain import io
from torchain.functions import ChainResults, chain_loss
from torchain_amd import synth
import recipe_fixture as rf
P = 32
HYPER = dict(l2_regularize=5e-5, leaky_hmm_coefficient=0.1, xent_regularize=0.1, kaldi_way=True)
fst = synth.random_den_fst(120, 4, P, seed=3)
d = pathlib.Path(tempfile.mkdtemp())
ntr = int(sys.argv[1]); lr = float(sys.argv[2]); mom = float(sys.argv[3]); epochs = int(sys.argv[4])
train = rf.write_learnable_set(d, fst, [20] * (2 * ntr // 3) + [14] * (ntr // 3), seed=100, name="train")
valid = rf.write_learnable_set(d, fst, [20] * 16 + [14] * 8, seed=900, name="valid")
den = io.DenominatorGraph(fst, P); den.prepare("cuda:0")
tr = io.RandExample(train, seed=1, batchsize=8); va = io.RandExample(valid, seed=1, batchsize=8)
torch.manual_seed(0)
model = rf.TwoLayerTdnn(P).cuda()
opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=mom)
def fwd(data):
    (f, iv), sup = data
    a, b = model(f.cuda(), iv.cuda())
    return chain_loss(a, den, sup, xent_input=b, **HYPER)
def validate(nograd):
    r = ChainResults(); va.reset()
    for data in va:
        if nograd:
            with torch.no_grad(): _, res = fwd(data)
        else:
            _, res = fwd(data)
        r.data += res.data
    return float(r.loss)
print("valid before: eval %.4f train-path %.4f" % (validate(True), validate(False)))
for e in range(epochs):
    tr.reset(); r = ChainResults(); ls = []
    for i, data in enumerate(tr, 1):
        loss, res = fwd(data); loss.backward()
        if i % 2 == 0: opt.step(); opt.zero_grad()
        r.data += res.data; ls.append(float(res.loss))
    print("epoch %d train %.4f (%s) valid eval %.4f train-path %.4f" % (e, float(r.loss), " ".join("%.2f" % v for v in ls), validate(True), validate(False)))
