"""Which end slot gives a window its pedestal (spectra.c:66-78: min(k[first], k[last])) on synthetic
line tables, by a numpy restatement of the recurrence per bin (Lorentz wings only: statistics, not
parity).  The relaxation of pedestal.h relies on long chains of first-slot minima being rare."""
import numpy as np, sys
sys.path.insert(0, ".")
from pylbl_amd import synthetic
def stats(f, P=98388., T=288.99, x=None):
    t = synthetic.line_table(f, 1., 5000.)
    surf = synthetic.surface_level()
    x = surf.vmr[f][0] if x is None else x
    p = P*9.86923e-6; pp = p*x; tf = 296./T
    nu = t.nu; nup = nu + p*t.delta_air
    gamma = (t.gamma_air*(p-pp) + t.gamma_self*pp)*tf**t.n_air
    S = t.sw*np.exp(t.elower*1.4387752*(T-296.)/(T*296.))*(1-np.exp(-1.4387752*nu/T))/(1-np.exp(-1.4387752*nu/296.))*1e-4
    v0, vn, cut = 1, 5001, 25
    b = np.floor(nup).astype(int)
    ncell = vn - v0
    A = np.zeros(ncell + 1)      # slots (ignore extra slot subtlety: treat top as slot ncell)
    order = np.argsort(b, kind="stable")   # approx: group by bin (ignores alternation)
    wins_e = wins_s = 0
    dep_chain = 0; longest = 0
    ub = np.unique(b)
    idx = {bb: np.where(b == bb)[0] for bb in ub}
    for bb in ub:
        fs = max(bb - cut - v0, 0); ls = min(bb + cut + 1 - v0, ncell)
        if fs > ncell - 1 or ls < 0: continue
        slots = np.arange(fs, ls + 1)
        vv = v0 + slots.astype(float)
        j = idx[bb]
        d = vv[None, :] - nup[j, None]
        G = (S[j, None]*gamma[j, None]/np.pi/(d*d + gamma[j, None]**2)).sum(0)
        A[fs:ls+1] += G
        ks, ke = A[fs], A[ls]
        if ks < ke:
            wins_s += 1; dep_chain += 1; longest = max(longest, dep_chain)
        else:
            wins_e += 1; dep_chain = 0
        A[fs:ls+1] -= min(ks, ke)
    print(f, "P", P, "bins", len(ub), "k_s wins", wins_s, "k_e wins", wins_e, "longest k_s streak", longest)
for f in ("H2O", "CO2"):
    stats(f)
stats("CO2", P=1000., T=230.)

def stats_table(t, label, P=98388., T=288.99, x=3.6e-4, v0=1, vn=5001, cut=25):
    p = P*9.86923e-6; pp = p*x; tf = 296./T
    nu = t.nu; nup = nu + p*t.delta_air
    gamma = (t.gamma_air*(p-pp) + t.gamma_self*pp)*tf**t.n_air
    S = t.sw*np.exp(t.elower*1.4387752*(T-296.)/(T*296.))*(1-np.exp(-1.4387752*nu/T))/(1-np.exp(-1.4387752*nu/296.))*1e-4
    b = np.floor(nup).astype(int)
    ncell = vn - v0
    A = np.zeros(ncell + 1)
    wins_e = wins_s = 0; dep = 0; longest = 0
    ub = np.unique(b)
    srt = np.argsort(b, kind="stable"); bs = b[srt]
    starts = np.searchsorted(bs, ub); ends = np.searchsorted(bs, ub, side="right")
    for bb, s0, e0 in zip(ub, starts, ends):
        fs = max(bb - cut - v0, 0); ls = min(bb + cut + 1 - v0, ncell)
        if fs > ncell - 1 or ls < 0: continue
        slots = np.arange(fs, ls + 1); vv = v0 + slots.astype(float)
        j = srt[s0:e0]
        d = vv[None, :] - nup[j, None]
        G = (S[j, None]*gamma[j, None]/np.pi/(d*d + gamma[j, None]**2)).sum(0)
        A[fs:ls+1] += G
        ks, ke = A[fs], A[ls]
        if ks < ke:
            wins_s += 1; dep += 1; longest = max(longest, dep)
        else:
            wins_e += 1; dep = 0
        A[fs:ls+1] -= min(ks, ke)
    print(label, "bins", len(ub), "k_s wins", wins_s, "k_e wins", wins_e, "longest k_s streak", longest)

stats_table(synthetic.banded_line_table("CO2", 1., 5000., num_lines=400000, bands=8, seed=42), "banded CO2")
stats_table(synthetic.line_table("CO2", 1., 5000., num_lines=300, seed=5), "sparse 300 lines")
stats_table(synthetic.line_table("CO2", 1., 5000., num_lines=5000, seed=6), "5000 lines")
stats_table(synthetic.line_table("CO2", 1., 5000., num_lines=20000, seed=7), "20000 lines")
