"""
Packs the published ENM parameter tables that springcraft ships as CSV (src/springcraft/data/*.csv, loaded by
forcefield.py:943-950) into ONE binary archive, springcraft_amd/data/enm_tables.npz.  Run in the build container
(the reference is not available on the GPU box).  These are literature constants, not code:
  miyazawa  Miyazawa & Jernigan, J Mol Biol 256, 623 (1996)            (20, 20)
  keskin    Keskin, Bahar, Jernigan, Badretdinov, Ptitsyn, Protein Sci 7, 2578 (1998)   (20, 20)
  s_enm_10, s_enm_13, d_enm, d_enm_edges, sd_enm   Dehouck & Mikhailov, PLoS Comput Biol 9, e1003209 (2013)
Amino-acid order: alphabetical by one-letter code (A C D E F G H I K L M N P Q R S T V W Y).
"""
import os
import sys

import numpy as np

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/springcraft/data"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "springcraft_amd", "data",
                   "enm_tables.npz")
names = ["miyazawa", "keskin", "s_enm_10", "s_enm_13", "d_enm", "d_enm_edges", "sd_enm"]
tables = {n: np.loadtxt(os.path.join(SRC, n + ".csv"), delimiter=",") for n in names}
for n, t in tables.items():
    print(n, t.shape)
np.savez_compressed(OUT, **tables)
print("wrote", OUT, os.path.getsize(OUT), "bytes")
