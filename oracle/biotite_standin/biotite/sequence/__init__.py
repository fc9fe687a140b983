"""Minimal ``biotite.sequence`` stand-in: the amino-acid alphabet order springcraft uses."""

_ONE = "ACDEFGHIKLMNPQRSTVWY" + "BZX*"
_THREE = {
    "A": "ALA", "C": "CYS", "D": "ASP", "E": "GLU", "F": "PHE", "G": "GLY", "H": "HIS",
    "I": "ILE", "K": "LYS", "L": "LEU", "M": "MET", "N": "ASN", "P": "PRO", "Q": "GLN",
    "R": "ARG", "S": "SER", "T": "THR", "V": "VAL", "W": "TRP", "Y": "TYR",
    "B": "ASX", "Z": "GLX", "X": "UNK", "*": " * ",
}


class _Alphabet:
    def get_symbols(self):
        return list(_ONE)


class ProteinSequence:
    alphabet = _Alphabet()

    @staticmethod
    def convert_letter_1to3(symbol):
        return _THREE[symbol.upper()]
