"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the reference's MT-CKD continuum
path, the checker for pylbl_amd's continuum kernels.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; the product never does.

What is restated (reference file:line):
  * number densities and the radiation term      pyLBL/mt_ckd/utils.py:16-59
  * the 16 band formulas                         water_vapor.py, carbon_dioxide.py, nitrogen.py,
                                                 oxygen.py, ozone.py (cited per band below)
  * interpolation to the user's grid, x100       utils.py:157-174 (numpy.interp, zero outside)

Pinned by the 16 known-answer values of the reference's own test (tests/test_mt_ckd.py:15-26:
the sum of every band's spectrum for the last level of the fixture atmosphere) evaluated on
the reference's coefficient file (tests/golden/mt_ckd_bands.npz is its conversion).

Layout: one table of bands; a band is (owner, arrays it reads, coarse grid, formula).  The
formula gets a dictionary of level scalars and numpy arrays and returns the coarse spectrum
[cm-1] of utils.py's ``Continuum.spectra``.
"""
import numpy as np

LOSCHMIDT = 2.6867775e19      # utils.py:7
P0 = 1013.25                  # utils.py:8  [mb]
C2 = 1.4387752                # utils.py:9  [cm K]
T0 = 296.                     # utils.py:10
T273 = 273.15                 # utils.py:11


def load_tables(path):
    """{name: (data, lower, upper, resolution)} from the .npz fixture."""
    tables = {}
    with np.load(path) as archive:
        for name in archive.files:
            if not name.endswith("__grid"):
                lower, upper, resolution = archive[name + "__grid"]
                tables[name] = (np.array(archive[name], dtype=np.float64), float(lower),
                                float(upper), float(resolution))
    return tables


def coarse_grid(lower, resolution, size):
    return np.asarray([lower + i*resolution for i in range(size)])    # utils.py:142-143


def radiation_term(w, temperature):
    """utils.py:45-59; its x <= 0.01 branch is always overwritten by the x <= 10 one."""
    x = w/(temperature/C2)
    small = np.where(x <= 0.01, 0.5*x*w, w)
    return np.where(x <= 10., w*(1. - np.exp(-x))/(1. + np.exp(-x)), small)


def dry_air(pressure, temperature, vmr):
    return LOSCHMIDT*(pressure/P0)*(T273/temperature)*(1. - vmr["H2O"])    # utils.py:31-42


def moist_air(pressure, temperature, vmr):
    return sum([dry_air(pressure, temperature, vmr)*x for x in vmr.values()])  # utils.py:16-28


def window(outer, inner):
    """Indices of a sub-table inside its parent (utils.py:62-80)."""
    if outer[3] != inner[3]:
        raise ValueError("grid and subgrid have different resolutions.")
    if outer[1] > inner[1] or outer[2] < inner[2]:
        raise ValueError("subgrid not contained in grid.")
    return int((inner[1] - outer[1])/outer[3]), int((inner[2] - outer[1])/outer[3])


# --- band set-up (what the reference does in the constructors) ---------------------------

def setup_h2o_self(t):
    return t["bs296"][1], t["bs296"][3], {"c296": t["bs296"][0], "c260": t["bs260"][0]}


def setup_h2o_foreign(t):
    base, fac = t["bfh2o"], t["xfac_rhu"]
    w = coarse_grid(base[1], base[3], base[0].size)
    lo, hi = window(base, fac)
    scale = np.zeros(base[0].size)                        # water_vapor.py:55-59
    scale[lo + 1:hi + 1] = fac[0][1:]
    scale[lo] = scale[lo + 1]
    tail = w[hi + 1:]                                     # water_vapor.py:60-69
    minus = (tail - 255.67)*(tail - 255.67)
    minus8 = np.power((tail - 255.67)/57.83, 8)
    plus = (tail + 255.67)*(tail + 255.67)
    plus8 = np.power((tail + 255.67)/57.83, 8)
    roll = np.power(tail/630., 8)
    scale[hi + 1:] = 1. + (0.06 - 0.42*((57600./(minus + 57600. + minus8)) +
                                        (57600./(plus + 57600. + plus8))))/(1. + 0.3*roll)
    return base[1], base[3], {"c": base[0], "scale": scale}


def setup_co2(t):
    base = t["bfco2"]
    exponent = np.ones(base[0].size)                      # carbon_dioxide.py:23-31
    lo, hi = window(base, t["tdep_bandhead"])
    exponent[lo:hi + 1] = t["tdep_bandhead"][0]
    chi = np.ones(base[0].size)
    lo, hi = window(base, t["x_factor_co2"])
    chi[lo:hi + 1] = t["x_factor_co2"][0]
    return base[1], base[3], {"c": base[0], "chi": chi, "exponent": exponent}


def plain(*names):
    def setup(t):
        first = t[names[0]]
        return first[1], first[3], {name: t[name][0] for name in names}
    return setup


def setup_o2_nir2(t):
    w = np.arange(9100., 11002., 2.)                      # oxygen.py:56-67
    out = np.zeros(w.size)
    for i, v in enumerate(w):
        d1, d2 = v - 9375., v - 9439.
        a1 = np.exp(d1/176.1) if d1 < 0. else 1.
        a2 = np.exp(d2/176.1) if d2 < 0. else 1.
        out[i] = 0.31831*(((1.166e-04*a1/58.96)/(1. + (d1/58.96)*(d1/58.96))) +
                          ((3.086e-05*a2/45.04)/(1. + (d2/45.04)*(d2/45.04))))*1.054/v
    return 9100., 2., {"c": out}


def setup_o2_herzberg(t):
    w = np.arange(36000., 100010., 10.)                   # oxygen.py:113-124
    out = np.zeros(w.size)
    for i, v in enumerate(w):
        if v <= 36000.:
            continue
        correction = ((40000. - v)/4000.)*7.917e-7 if v <= 40000. else 0.
        ratio = v/48811.0
        out[i] = 6.884e-4*ratio*np.exp(-69.738*np.power(np.log(ratio), 2)) - correction
    return 36000., 10., {"c": out}


# --- band formulas (what the reference does in Continuum.spectra) ------------------------

def h2o_self(s, a):                                       # water_vapor.py:23-32
    nh2o = s["dry"]*s["vmr"]["H2O"]
    return nh2o*(nh2o/s["air"])*(s["p"]/P0)*(T0/s["t"])*1.e-20*s["rad"] * \
        a["c296"]*np.power(a["c260"]/a["c296"], (s["t"] - T0)/(260. - T0))


def h2o_foreign(s, a):                                    # water_vapor.py:71-78
    nh2o = s["dry"]*s["vmr"]["H2O"]
    return (1. - (nh2o/s["air"]))*(s["p"]/P0)*(T0/s["t"])*1.e-20*nh2o*s["rad"]*a["scale"]*a["c"]


def co2(s, a):                                            # carbon_dioxide.py:33-39
    n = s["dry"]*s["vmr"]["CO2"]
    return n*1.e-20*(s["p"]/P0)*(T0/s["t"])*s["rad"]*a["chi"] * \
        np.power(s["t"]/246., a["exponent"])*a["c"]


def n2_tau(s):
    return (s["dry"]*s["vmr"]["N2"]/LOSCHMIDT)*(s["p"]/P0)*(T273/s["t"])


def n2_rotation(s, a):                                    # nitrogen.py:19-30
    f = (s["t"] - T0)/(220. - T0)
    c = a["ct_296"]*np.power(a["ct_220"]/a["ct_296"], f)
    sf = a["sf_296"]*np.power(a["sf_220"]/a["sf_296"], f)
    fo2 = (sf - 1.)*s["vmr"]["N2"]/s["vmr"]["O2"]
    return n2_tau(s)*s["rad"]*c*(s["vmr"]["N2"] + fo2*s["vmr"]["O2"] + s["vmr"]["H2O"])


def n2_fundamental(s, a):                                 # nitrogen.py:40-54
    xt = (1./s["t"] - 1./272.)/(1./228. - 1./272.)
    c0 = np.zeros(a["xn2_272"].size)
    c0[1:-1] = a["xn2_272"][1:-1]*np.power(a["xn2_228"][1:-1]/a["xn2_272"][1:-1], xt)
    c0 = c0/s["w"]
    c1 = (1.294 - 0.4545*s["t"]/T0)*c0
    c2 = (9./7.)*a["a_h2o"]*c0
    return n2_tau(s)*s["rad"]*(c0*s["vmr"]["N2"] + s["vmr"]["O2"]*c1 + s["vmr"]["H2O"]*c2)


def n2_overtone(s, a):                                    # nitrogen.py:63-68
    mix = s["vmr"]["N2"] + s["vmr"]["O2"] + s["vmr"]["H2O"]
    return n2_tau(s)*mix*s["rad"]*a["xn2"]/s["w"]


def o2_fundamental(s, a):                                 # oxygen.py:22-30
    no2 = s["dry"]*s["vmr"]["O2"]
    tau = no2*1.e-20*(s["p"]/P0)*(T273/s["t"])
    return tau*s["rad"]*(1.e20/LOSCHMIDT)*a["o2_f"]*np.exp(a["o2_t"]*((1./T0) - (1./s["t"])))/s["w"]


def o2_nir(s, a):                                         # oxygen.py:39-47
    no2 = s["dry"]*s["vmr"]["O2"]
    tau = (no2/LOSCHMIDT)*(s["p"]/P0)*(T273/s["t"]) * \
        ((1./0.446)*s["vmr"]["O2"] + (0.3/0.446)*s["vmr"]["N2"] + s["vmr"]["H2O"])
    return tau*s["rad"]*a["o2_inf1"]/s["w"]


def o2_nir2(s, a):                                        # oxygen.py:69-74
    no2 = s["dry"]*s["vmr"]["O2"]
    adj = (no2/s["air"])*(1./s["vmr"]["O2"])*no2*1.e-20*(s["p"]/P0)*(T0/s["t"])
    return adj*s["rad"]*a["c"]


def o2_nir3(s, a):                                        # oxygen.py:90-94
    no2 = s["dry"]*s["vmr"]["O2"]
    return (no2/LOSCHMIDT)*(s["p"]/P0)*(T273/s["t"])*s["rad"]*a["o2_inf3"]/s["w"]


def o2_visible(s, a):                                     # oxygen.py:105-111
    no2 = s["dry"]*s["vmr"]["O2"]
    adj = (no2/s["air"])*no2*1.e-20*(s["p"]/P0)*(T273/s["t"])
    factor = 1./(LOSCHMIDT*1.e-20*(55.*T273/T0)*(55.*T273/T0)*89.5)
    return adj*s["rad"]*factor*a["o2_invis"]/s["w"]


def o2_herzberg(s, a):                                    # oxygen.py:126-130
    no2 = s["dry"]*s["vmr"]["O2"]
    return 1.e-20*no2*s["rad"]*(1. + 0.83*(s["p"]/P0)*(T273/s["t"]))*a["c"]/s["w"]


def o2_uv(s, a):                                          # oxygen.py:138-141
    return 1.e-20*s["dry"]*s["vmr"]["O2"]*s["rad"]*a["o2_infuv"]/s["w"]


def o3_chappuis(s, a):                                    # ozone.py:23-28
    dt = s["t"] - T273
    return 1.e-20*s["dry"]*s["vmr"]["O3"]*s["rad"]*(a["x_o3"] + a["y_o3"]*dt +
                                                     a["z_o3"]*dt*dt)/s["w"]


def o3_hartley(s, a):                                     # ozone.py:46-52
    dt = s["t"] - T273
    return 1.e-20*s["dry"]*s["vmr"]["O3"]*s["rad"]*(a["o3_hh0"]/s["w"]) * \
        (1. + a["o3_hh1"]*dt + a["o3_hh2"]*dt*dt)


def o3_uv(s, a):                                          # ozone.py:68-71
    return s["dry"]*s["vmr"]["O3"]*s["rad"]*a["o3_huv"]/s["w"]


# owner (key of the reference's continua dictionary, setup.py:47-54) -> its bands in order.
BANDS = {
    "H2OSelf": [(setup_h2o_self, h2o_self)],
    "H2OForeign": [(setup_h2o_foreign, h2o_foreign)],
    "CO2": [(setup_co2, co2)],
    "N2": [(plain("ct_296", "ct_220", "sf_296", "sf_220"), n2_rotation),
           (plain("xn2_272", "xn2_228", "a_h2o"), n2_fundamental),
           (plain("xn2"), n2_overtone)],
    "O2": [(plain("o2_f", "o2_t"), o2_fundamental), (plain("o2_inf1"), o2_nir),
           (setup_o2_nir2, o2_nir2), (plain("o2_inf3"), o2_nir3),
           (plain("o2_invis"), o2_visible), (setup_o2_herzberg, o2_herzberg),
           (plain("o2_infuv"), o2_uv)],
    "O3": [(plain("x_o3", "y_o3", "z_o3"), o3_chappuis),
           (plain("o3_hh0", "o3_hh1", "o3_hh2"), o3_hartley), (plain("o3_huv"), o3_uv)],
}


class Continuum(object):
    """All bands of one owner, set up once."""
    def __init__(self, owner, tables):
        self.bands = []
        for setup, formula in BANDS[owner]:
            lower, resolution, arrays = setup(tables)
            size = next(iter(arrays.values())).size
            self.bands.append((coarse_grid(lower, resolution, size), arrays, formula))

    def band_spectra(self, temperature, pressure_mb, vmr):
        """List of coarse spectra [cm-1], one per band (``Continuum.spectra``)."""
        out = []
        for w, arrays, formula in self.bands:
            scalars = {"t": temperature, "p": pressure_mb, "vmr": vmr, "w": w,
                       "dry": dry_air(pressure_mb, temperature, vmr),
                       "air": moist_air(pressure_mb, temperature, vmr),
                       "rad": radiation_term(w, temperature)}
            with np.errstate(divide="ignore", invalid="ignore"):
                out.append(formula(scalars, arrays))
        return out

    def spectra(self, temperature, pressure_pa, vmr, grid):
        """``BandedContinuum.spectra`` (utils.py:157-174): extinction [m-1] on ``grid``."""
        total = np.zeros(grid.size)
        coarse = self.band_spectra(temperature, pressure_pa*0.01, vmr)
        for (w, _, _), values in zip(self.bands, coarse):
            total += np.interp(grid, w, values, left=0., right=0.)*100.
        return total
