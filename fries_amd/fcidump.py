"""FCIDUMP input for the FRI engine: parser (host logic) and the seeded synthetic
generator that stands in for the integral files the reference checkout lacks.

Parser rules follow the reference's ``parse_fcidump`` (FRIES/io_utils.cpp:241-318):
line 1 holds NORB/NELEC/MS2, line 2 ORBSYM, lines 3-4 are skipped (ISYM, &END), every
further record is ``value i j k l`` (1-based, chemist order).  Irrep labels are mapped
to the engine's internal labels like ``convert_symm`` (io_utils.cpp:189-239).

Two-electron integrals are kept 8-fold packed exactly like the reference's
``SymmERIs`` (FRIES/ndarr.hpp:206-244): pair index p(i<=j) = j(j+1)/2 + i, element
index = p2(p2+1)/2 + p1 for p1 <= p2.
"""
from __future__ import annotations

import dataclasses
import numpy as np

# FRIES/io_utils.cpp:190-238
_IRREP_MAPS = {
    "d2h": [0, 7, 6, 1, 5, 2, 3, 4],
    "c2v": [0, 2, 3, 1],
    "c2h": [0, 2, 3, 1],
    "d2": [0, 3, 2, 1],
    "cs": [0, 1], "c2": [0, 1], "ci": [0, 1], "c1": [0, 1],
}

# Irrep labels (internal numbering) of the unfrozen orbitals of the reference's example
# systems: Input_Data/{N2_ccpvdz,H2O_ccpvdz,Neon_augccpvdz}/symm.txt minus frozen cores.
_SHAPES = {
    # name: (n_elec, point group, irreps)
    "N2": (10, "D2h", [0, 5, 0, 6, 7, 2, 3, 5, 0, 6, 7, 0, 2, 3, 5, 5, 0, 1, 6, 7, 4, 5, 0, 2, 3, 5]),
    "H2O": (10, "C2v", [0, 0, 3, 0, 2, 0, 3, 3, 0, 0, 2, 3, 1, 0, 2, 3, 0, 3, 0, 2, 1, 0, 0, 3]),
    "Ne": (8, "D2h", [0, 5, 6, 7, 0, 5, 6, 7, 0, 0, 1, 2, 3, 5, 6, 7, 0, 0, 0, 1, 2, 3]),
    # edge cases for the tests (not reference systems): the largest index one 64-bit word holds (32 spatial orbitals, every bit
    # of the determinant used), in two irreps; and the smallest closed shell with a choice to make (2 electrons in 4 orbitals)
    "MAX32": (6, "Cs", [0, 1] * 16),
    "MIN4": (2, "C1", [0, 0, 0, 0]),
}


@dataclasses.dataclass
class MolInput:
    n_orb: int
    n_elec: int
    irreps: np.ndarray      # uint8[n_orb], internal labels
    h_core: np.ndarray      # float64[n_orb, n_orb]
    eris: np.ndarray        # float64[packed_len(n_orb)]
    core_en: float = 0.0
    point_group: str = "C1"


def packed_len(n_orb: int) -> int:
    p = n_orb * (n_orb + 1) // 2
    return p * (p + 1) // 2


def _pair(i: int, j: int) -> int:
    return (j * (j + 1)) // 2 + i if i <= j else (i * (i + 1)) // 2 + j


def eri_index(i: int, j: int, k: int, l: int) -> int:
    """Packed position of the chemist-order integral (ij|kl)."""
    p1, p2 = _pair(i, j), _pair(k, l)
    if p1 > p2:
        p1, p2 = p2, p1
    return (p2 * (p2 + 1)) // 2 + p1


def convert_symm(labels, point_group: str) -> np.ndarray:
    pg = point_group.lower()
    if pg not in _IRREP_MAPS:
        raise RuntimeError(f"Point group {point_group} not recognized")
    mp = _IRREP_MAPS[pg]
    out = np.empty(len(labels), dtype=np.uint8)
    for n, lab in enumerate(labels):
        if lab > len(mp) or lab < 1:
            raise RuntimeError(f"irrep index {lab} read from the FCIDUMP file exceeds the maximum allowed irrep index ({len(mp)}) for point group {point_group}")
        out[n] = mp[lab - 1]
    return out


def parse_fcidump(path: str, point_group: str = "C1") -> MolInput:
    with open(path) as f:
        line1 = f.readline()
        line2 = f.readline()
        f.readline()
        f.readline()
        body = f.read().split()

    def _field(line, key):
        pos = line.find(key)
        end = line.find(",", pos)
        return int(line[pos + len(key):end if end >= 0 else None])

    n_orb = _field(line1, "NORB=")
    n_elec = _field(line1, "NELEC=")
    if _field(line1, "MS2=") != 0:
        raise RuntimeError("MS2 is not zero in FCIDUMP file.")
    labels = []
    for tok in line2[line2.find("ORBSYM=") + 7:].split(","):
        tok = tok.strip()
        if tok:
            try:
                labels.append(int(tok))
            except ValueError:
                pass
    if len(labels) != n_orb:
        raise RuntimeError("Number of irrep labels read in after ORBSYM in FCIDUMP file does not equal number of orbitals")
    irreps = convert_symm(labels, point_group)
    h = np.zeros((n_orb, n_orb))
    eris = np.zeros(packed_len(n_orb))
    core = 0.0
    for r in range(0, len(body) - 4, 5):
        val = float(body[r])
        a, b, c, d = (int(x) for x in body[r + 1:r + 5])
        if a == 0 and b == 0 and c == 0 and d == 0:
            core = val
        elif b == 0 and c == 0 and d == 0:
            continue
        elif c == 0 and d == 0:
            h[a - 1, b - 1] = h[b - 1, a - 1] = val
        else:
            # the reference stores through chemist_ordered(d-1, c-1, b-1, a-1) with d<=c, b<=a
            # and pair(d,c) <= pair(b,a); eri_index is symmetric so any order lands the same
            eris[eri_index(a - 1, b - 1, c - 1, d - 1)] = val
    return MolInput(n_orb, n_elec, irreps, h, eris, core, point_group)


def synthetic(shape: str = "N2", seed: int = 12345) -> MolInput:
    """Seeded FCIDUMP-shaped integrals: only symmetry-allowed
    (ij|kl) are non-zero; Coulomb-like diagonal, small random remainder."""
    n_elec, pg, irr = _SHAPES[shape]
    irr = np.asarray(irr, dtype=np.uint8)
    n = len(irr)
    rng = np.random.RandomState(seed)
    h = np.zeros((n, n))
    for i in range(n):
        h[i, i] = -8 + 0.45 * i
        for j in range(i):
            if irr[i] == irr[j]:
                h[i, j] = h[j, i] = 0.02 * (rng.random_sample() - 0.5)
    eris = np.zeros(packed_len(n))
    for i in range(n):
        for j in range(i + 1):
            for k in range(i + 1):
                for l in range(k + 1):
                    if _pair(j, i) < _pair(l, k):
                        continue
                    if irr[i] ^ irr[j] ^ irr[k] ^ irr[l]:
                        continue
                    if i == j and k == l:
                        val = 0.6 / (1 + 0.15 * abs(i - k))
                    else:
                        val = 0.05 * (rng.random_sample() - 0.5) / (1 + 0.2 * (abs(i - j) + abs(k - l)))
                    eris[eri_index(i, j, k, l)] = val
    return MolInput(n, n_elec, irr, h, eris, 0.0, pg)


def write_fcidump(path: str, mol: MolInput) -> None:
    mp = _IRREP_MAPS[mol.point_group.lower()]
    orbsym = [mp.index(int(x)) + 1 for x in mol.irreps]
    n = mol.n_orb
    with open(path, "w") as f:
        f.write(f" &FCI NORB={n},NELEC={mol.n_elec},MS2=0,\n")
        f.write("  ORBSYM=" + ",".join(str(x) for x in orbsym) + ",\n")
        f.write("  ISYM=1,\n &END\n")
        for i in range(n):
            for j in range(i + 1):
                for k in range(i + 1):
                    for l in range(k + 1):
                        if _pair(j, i) < _pair(l, k):
                            continue
                        v = mol.eris[eri_index(i, j, k, l)]
                        if v != 0.0:
                            f.write(f"{float(v)!r} {i + 1} {j + 1} {k + 1} {l + 1}\n")
        for i in range(n):
            for j in range(i + 1):
                if mol.h_core[i, j] != 0.0:
                    f.write(f"{float(mol.h_core[i, j])!r} {i + 1} {j + 1} 0 0\n")
        f.write(f"{float(mol.core_en)!r} 0 0 0 0\n")


def write_hf_dir(path: str, mol, eps: float = 0.01, hf_energy: float = 0.0) -> None:
    """The legacy HF-output directory the reference's frifull_mol / frimulti_mol read (parse_hf_input, FRIES/io_utils.cpp:98-187), no
    frozen orbitals: sys_params.txt, symm.txt (the library's irrep labels), hcore.txt, eris.txt with eris[i][j][k][l] = <ij|kl> =
    (ik|jl), 17 significant digits so that the text round-trips bit for bit.  `path` ends with a slash."""
    n = mol.n_orb
    with open(path + "sys_params.txt", "w") as f:
        f.write("n_elec\n%d\nn_frozen\n0\nn_orb\n%d\neps\n%r\nhf_energy\n%r\n" % (mol.n_elec, n, float(eps), float(hf_energy)))
    with open(path + "symm.txt", "w") as f:
        f.write(",".join(str(int(x)) for x in mol.irreps) + "\n")
    h = np.asarray(mol.h_core, dtype=np.float64).reshape(n, n)
    with open(path + "hcore.txt", "w") as f:
        for i in range(n):
            f.write(",".join("%.17g" % x for x in h[i]) + "\n")
    tri = lambda a, b: np.where(a <= b, b * (b + 1) // 2 + a, a * (a + 1) // 2 + b)
    ii, jj, kk, ll = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    p1, p2 = tri(ii, kk), tri(jj, ll)                    # <ij|kl> = (ik|jl)
    e4 = np.asarray(mol.eris, dtype=np.float64)[tri(p1, p2)].reshape(n * n * n, n)
    with open(path + "eris.txt", "w") as f:
        for row in e4:
            f.write(",".join("%.17g" % x for x in row) + "\n")
