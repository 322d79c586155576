import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from professad_amd.engine import Engine
from professad_amd import functionals as F
from professad_amd.ions import ion_ion, ionic_potential, recpot_table
from professad_amd.optimize import EV_PER_HA, optimize_density
g = np.load('tests/golden/ions.npz')
tab = recpot_table(g['recpot_raw'], float(g['recpot_kmax']))
box = 4.050 / 0.529177210903 * np.array([[0.5, 0.5, 0.0], [0.0, 0.5, 0.5], [0.5, 0.0, 0.5]])
frac = np.zeros((1, 3))
eng = Engine((18, 18, 18), 'cuda:0').set_cell(torch.as_tensor(box))
vext = ionic_potential(eng, box, [(frac, tab)])
eng.set_terms(F.NativeTerms(['ion_electron', 'hartree', 'wt', 'pbe']).names)
res = optimize_density(eng, float(tab[2]), vext, volume=abs(np.linalg.det(box)), ntol=1e-7)
E_ii, _, _ = ion_ion(eng, box, frac, [float(tab[2])])
print('E_total_eV %.9f  (PROFESS4 -57.183329402)  iterations %d  E_ii_Ha %.9f' % ((res['E_Ha'] + E_ii) * EV_PER_HA, res['iterations'], E_ii))
