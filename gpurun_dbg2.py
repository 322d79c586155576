import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import closed_form as cf
from professad_amd.engine import Engine
d = np.load('tests/golden/cfg1_fccAl_32.npz')
box, vext, n_elec = d['box'], d['vext'], float(d['n_elec'])
chi = np.load('gpurun_chi1.npy')
dev='cuda:0'; t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.double, device=dev)
ev = cf.Evaluator(cf.Grid(box, chi.shape))
eng = Engine(chi.shape, dev).set_cell(torch.as_tensor(box))
for names, onames in ((['ion_electron'],['ion_electron']),(['hartree'],['hartree']),(['tf'],['tf']),(['vw'],['vw']),(['lda_x'],['lda_x']),(['pz_c'],['pz_c']),(['ion_electron','hartree','tf','vw','lda_x','pz_c'],)*2):
    eng.set_terms(names)
    for mode in (0,1):
        eng.set_option(0, mode)
        E, mu, g = eng.energy_grad_chi(t(chi), n_elec, t(vext))
        Eo, go, muo = ev.closure(onames, chi, n_elec, vext)
        print(names, 'mode', mode, 'E', sum(E.values()), 'oracle', Eo, 'dE', sum(E.values())-Eo, 'dg', np.abs(g.cpu().numpy()-go).max()/np.abs(go).max(), flush=True)
