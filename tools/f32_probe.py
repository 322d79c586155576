"""fp32 build against the fp64 build on the same inputs: FFT round trip, energies, gradient, time per evaluation.
usage: python tools/f32_probe.py [N ...]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from professad_amd import synth  # noqa: E402
from professad_amd.engine import Engine  # noqa: E402

CFG = {'cfg2': ['ion_electron', 'hartree', 'tf', 'vw', 'wt_nl', 'lda_x', 'pz_c'],
       'cfg3': ['ion_electron', 'hartree', 'tf', 'vw', 'wgc99_nl', 'pbe_x', 'pbe_c']}


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [32, 64, 256]
    dev = 'cuda:0'
    for n in sizes:
        shape = (n, n, n)
        box = torch.as_tensor(synth.cubic_cell(n))
        den = synth.smooth_density(shape, seed=3)
        vext = synth.random_potential(shape, seed=4)
        chi = np.sqrt(den)
        nel = float(round(den.mean() * abs(np.linalg.det(box.numpy()))))
        e64 = Engine(shape, dev).set_cell(box)
        e32 = Engine(shape, dev, dtype=torch.float32).set_cell(box)
        t64 = lambda a: torch.as_tensor(a, dtype=torch.double, device=dev)  # noqa: E731
        t32 = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)  # noqa: E731
        x = t32(den)
        sp = e32.rfftn(x)
        ref = torch.fft.rfftn(t64(den).to(torch.float32).to(torch.double))
        out = {'n': n, 'rfftn_rel': float((sp.to(torch.complex128) - ref).abs().max() / ref.abs().max()),
               'roundtrip_rel': float((e32.irfftn(sp) - x).abs().max() / x.abs().max())}
        for cfg, names in CFG.items():
            e64.set_terms(names)
            e32.set_terms(names)
            Ea, mua, ga = e64.energy_grad_chi(t64(chi), nel, t64(vext))
            Eb, mub, gb = e32.energy_grad_chi(t32(chi), nel, t32(vext))
            tot_a, tot_b = sum(Ea.values()), sum(Eb.values())
            out[cfg] = {'E64': tot_a, 'E32': tot_b, 'dE_rel': abs(tot_a - tot_b) / abs(tot_a),
                        'term_rel_max': max(abs(Ea[k] - Eb[k]) / max(abs(Ea[k]), 1e-30) for k in Ea if Ea[k] != 0.0),
                        'terms_rel': {k: round(abs(Ea[k] - Eb[k]) / abs(Ea[k]), 9) for k in Ea if Ea[k] != 0.0},
                        'mu64': mua, 'mu32': mub,
                        'grad_rel': float((gb.double() - ga).abs().max() / ga.abs().max()), 'fast32': e32.fast_path}
            for eng, tt, key in ((e64, t64, 'ms64'), (e32, t32, 'ms32')):
                c, v = tt(chi), tt(vext)
                for _ in range(3):
                    eng.energy_grad_chi(c, nel, v)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    eng.energy_grad_chi(c, nel, v)
                torch.cuda.synchronize()
                out[cfg][key] = (time.perf_counter() - t0) / 10 * 1e3
        print(json.dumps(out), flush=True)
        e64.close()
        e32.close()


if __name__ == '__main__':
    main()
