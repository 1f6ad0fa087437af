"""hvcoord_t / hvcoord_init (reference src/share/hybvcoord_mod.F90:18-171): hybrid vertical coordinate read from the
ascii files the reference's namelists name (test/dcmip1-1/dcmip1-1.nl: vcoord/acme-72{m,i}.ascii).  The two data
files are shipped as data under transport_se_amd/data/vcoord/."""
import os

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "vcoord")
P0 = 100000.0  # physical_constants.F90:26


class HvCoord:
    def __init__(self, vfile_mid=None, vfile_int=None):
        vfile_mid = vfile_mid or os.path.join(DATA, "acme-72m.ascii")
        vfile_int = vfile_int or os.path.join(DATA, "acme-72i.ascii")
        self.hyai, self.hybi = self._read(vfile_int)
        self.hyam, self.hybm = self._read(vfile_mid)
        if self.hyai.size != 73 or self.hyam.size != 72:
            raise ValueError("hyai input file and plevp do not match")  # hybvcoord_mod.F90:79-82
        self.ps0 = P0
        self.etam = self.hyam + self.hybm   # :170-171
        self.etai = self.hyai + self.hybi

    @staticmethod
    def _read(path):
        toks = []
        with open(path) as f:
            for line in f:
                line = line.split("!")[0].strip()
                if line:
                    toks += line.split()
        n = int(toks[0])
        a = np.array(toks[1:1 + n], dtype=np.float64)
        if int(toks[1 + n]) != n:
            raise ValueError("malformed vertical coordinate file " + path)
        b = np.array(toks[2 + n:2 + 2 * n], dtype=np.float64)
        return a, b
