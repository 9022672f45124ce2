"""A stand-in for the few h5py calls the reference's post-processing scripts make (scripts/FluidHDF5toXMF.py,
scripts/CellHDF5toXMF.py read file attributes and dataset shapes only), answered from `h5dump -A`: h5py is not in the image,
the HDF5 command-line tools are.  Test tooling; put this directory on PYTHONPATH to run those scripts on files written by
hemocell_amd/compat/hdf5_output.h."""
import re
import subprocess

import numpy as np

H5DUMP = "/opt/conda/bin/h5dump"


class _Dataset:
    def __init__(self, shape, dtype):
        self.shape, self.dtype = shape, dtype


class File:
    def __init__(self, name, mode="r"):
        out = subprocess.run([H5DUMP, "-A", name], capture_output=True, text=True, check=True).stdout
        self.attrs, self._sets = {}, {}
        for m in re.finditer(r'ATTRIBUTE "([^"]+)" \{\s*DATATYPE\s+(\S+)\s*DATASPACE\s+SIMPLE \{ \( ([^)]*) \)[^}]*\}\s*DATA \{(.*?)\n\s*\}\s*\}', out, re.S):
            vals = [float(v) for v in re.sub(r"\(\d+(,\d+)*\):", " ", m.group(4)).replace(",", " ").split()]
            self.attrs[m.group(1)] = np.array(vals, dtype=np.float64 if "F" in m.group(2) else np.int64)
        for m in re.finditer(r'DATASET "([^"]+)" \{\s*DATATYPE\s+(\S+)\s*DATASPACE\s+SIMPLE \{ \( ([^)]*) \)', out):
            self._sets[m.group(1)] = _Dataset(tuple(int(v) for v in m.group(3).split(",")), m.group(2))

    def items(self):
        return list(self._sets.items())

    def keys(self):
        return list(self._sets.keys())

    def __getitem__(self, k):
        return self._sets[k]

    def __contains__(self, k):
        return k in self._sets

    def close(self):
        pass
