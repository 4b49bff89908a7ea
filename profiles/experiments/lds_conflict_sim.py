"""Bank-conflict model of conv_dma_kernel's LDS reads (no GPU needed): replays the kernel's address arithmetic for one
workgroup of a layer, as planned by jaf_conv2d_plan_packed_ex, and counts the LDS-array cycles of every ds_read_b128 with
the lane groups and the bank rule of MI355X_MICROARCH.md (LDS section): a wave64 ds_read_b128 is served in four groups of
16 lanes -- {0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63} -- one cycle per group when the
16 lanes hit 64 distinct dword banks (bank = (addr/4) mod 64), one more cycle per extra distinct address on a busy bank.

  python profiles/experiments/lds_conflict_sim.py            # the 24-part / ConvLSTM layers of the stage-4 step

Prints, per layer: the plan, conflict cycles / conflict-free cycles of the patch (B operand) reads and of all reads
(A operand reads are 64 consecutive 16-byte items: conflict-free)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jafpro_amd import _lib                      # noqa: E402
from jafpro_amd._lib import ConvDesc, ConvPlan   # noqa: E402

GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def read_cycles(addrs):
    """LDS-array cycles of one ds_read_b128 given the 64 lanes' byte addresses."""
    cyc = 0
    for g in GROUPS:
        banks = {}
        for l in g:
            a = addrs[l]
            for k in range(4):
                banks.setdefault(((a >> 2) + k) & 63, set()).add(a)
        cyc += max(len(v) for v in banks.values())
    return cyc


def desc(N, G, Cin, Cout, H, W, K, stride=1, dil=1):
    d = ConvDesc()
    pad = (K - 1) // 2
    if dil == 2:                                   # data gradient of a stride-2 layer: zero-dilated input, pad K-1-p
        OH, OW = 2 * H, 2 * W
        pad = K - 1 - pad
    else:
        OH, OW = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    d.N, d.G, d.Cin, d.Cout, d.H, d.W, d.OH, d.OW = N, G, Cin, Cout, H, W, OH, OW
    d.KH = d.KW = K
    d.stride, d.pad_t, d.pad_l, d.dil_in, d.nsrc = stride, pad, pad, dil, 1
    d.src_c[0], d.src_ctot[0], d.src_coff[0], d.src_gstride[0] = Cin, G * Cin, 0, Cin
    d.w_cin_tot, d.w_cin_off, d.out_ctot, d.out_coff, d.act, d.slope, d.precision = Cin, 0, G * Cout, 0, 0, 0.0, 1
    return d


def simulate(d, lstm=0, flags=0, swz=None):
    L = _lib.lib()
    p = ConvPlan()
    rc = L.jaf_conv2d_plan_packed_ex(ctypes.byref(d), lstm, flags, ctypes.byref(p))
    assert rc == 0, rc
    NT, MT, NG = p.NT, p.MT, p.NG
    lg = (2 if NT == 4 else (1 if NT == 2 else 0)) if p.ilv else 0
    cmask = (1 << lg) - 1
    PWp, PWq = p.PWp, p.PWp >> lg
    taps = d.KH * d.KW
    tot = free = 0
    for wave in range(4):
        boff = {}
        for nt in range(NT):
            for li in range(16):
                pp = (wave * 16 * NT + li * NT + nt) if p.ilv else ((wave * NT + nt) * 16 + li)
                oy, oxr = pp // p.TWIN, pp % p.TWIN
                boff[(nt, li)] = ((oy * d.stride * PWp + (oxr >> lg) * d.stride) * 16)
        for last in ((0, 1) if p.nchunks > 1 and p.ng_last != NG else ((1,) if p.nchunks == 1 else (0,))):
            ngc = p.ng_last if last else NG
            nst = p.nsteps_last if last else p.nsteps
            for st in range(nst):
                for nt in range(NT):
                    addrs = []
                    for lane in range(64):
                        q, li = lane >> 4, lane & 15
                        s = 4 * st + q
                        v = 0
                        if s < taps * ngc:
                            tap, grp = s // ngc, s % ngc
                            ky, kx = tap // d.KW, tap % d.KW
                            xk = (d.stride * nt if p.ilv else 0) + kx
                            v = grp * p.plane + (ky * PWp + (xk & cmask) * PWq + (xk >> lg)) * 16
                        addrs.append(v + boff[(nt, li)])
                    tot += read_cycles(addrs)
                    free += 4
                tot_a = 4 * MT
                tot += tot_a
                free += tot_a
    return p, tot, free


LAYERS = [  # (label, N, G, Cin, Cout, H, W, K, stride, dil, lstm)
    ("cell 24->48 @200", 8, 24, 24, 48, 200, 200, 3, 1, 1, 1),
    ("cell 12->48 @200", 8, 24, 12, 48, 200, 200, 3, 1, 1, 1),
    ("cell 48->96 @100", 8, 24, 48, 96, 100, 100, 3, 1, 1, 1),
    ("cell 24->96 @100", 8, 24, 24, 96, 100, 100, 3, 1, 1, 1),
    ("cell 48->96 @50", 8, 24, 48, 96, 50, 50, 3, 1, 1, 1),
    ("cell 96->192 @25", 8, 24, 96, 192, 25, 25, 3, 1, 1, 1),
    ("cell 192->384 @13", 8, 24, 192, 384, 13, 13, 3, 1, 1, 1),
    ("dgrad 48->24 @200", 8, 24, 48, 24, 200, 200, 3, 1, 1, 0),
    ("dgrad 96->48 @100", 8, 24, 96, 48, 100, 100, 3, 1, 1, 0),
    ("enc1 3->12 k5 @200", 32, 24, 3, 12, 200, 200, 5, 1, 1, 0),
    ("enc2 12->24 s2 @200", 32, 24, 12, 24, 200, 200, 3, 2, 1, 0),
    ("enc3 24->24 @100", 32, 24, 24, 24, 100, 100, 3, 1, 1, 0),
    ("enc4 24->24 s2 @100", 32, 24, 24, 24, 100, 100, 3, 2, 1, 0),
    ("d(enc2) 24->12 @100 dil2", 32, 24, 24, 12, 100, 100, 3, 1, 2, 0),
    ("d(enc4) 24->24 @50 dil2", 32, 24, 24, 24, 50, 50, 3, 1, 2, 0),
    ("dec 36->12 @200", 8, 24, 36, 12, 200, 200, 3, 1, 1, 0),
    ("dec 12->12 @200", 8, 24, 12, 12, 200, 200, 3, 1, 1, 0),
    ("dec 12->3 k5 @200", 8, 24, 12, 3, 200, 200, 5, 1, 1, 0),
    ("dec 6->12 @200", 8, 24, 6, 12, 200, 200, 3, 1, 1, 0),
    ("dec 12->24 @200", 8, 24, 12, 24, 200, 200, 3, 1, 1, 0),
    ("dgrad 48->12 @200", 8, 24, 48, 12, 200, 200, 3, 1, 1, 0),
    ("dec 72->24 @100", 8, 24, 72, 24, 100, 100, 3, 1, 1, 0),
    ("dgrad 96->24 @100", 8, 24, 96, 24, 100, 100, 3, 1, 1, 0),
    ("CRN 256->256 @256", 8, 1, 256, 256, 256, 256, 3, 1, 1, 0),
    ("CRN 64->64 @256", 8, 1, 64, 64, 256, 256, 3, 1, 1, 0),
    ("D 6->32 s2 @256", 8, 1, 6, 32, 256, 256, 3, 2, 1, 0),
]

if __name__ == "__main__":
    print("%-28s %-44s %10s %10s" % ("layer", "plan", "cycles", "conflict/free"))
    for lab, N, G, Cin, Cout, H, W, K, s, dil, lstm in LAYERS:
        d = desc(N, G, Cin, Cout, H, W, K, s, dil)
        p, tot, free = simulate(d, lstm)
        plan = "MT%d NT%d NG%d TW%d PW%dp%d PH%d ilv%d ch%d st%d/%d" % (p.MT, p.NT, p.NG, p.TWIN, p.PW, p.PWp, p.PH, p.ilv, p.nchunks, p.nsteps, p.nsteps_last)
        print("%-28s %-44s %10d %10.2f" % (lab, plan, tot, (tot - free) / free))
