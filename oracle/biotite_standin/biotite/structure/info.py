"""``biotite.structure.info.mass`` stand-in: not needed for golden generation."""


def mass(item, is_residue=None):
    raise NotImplementedError("residue masses are not part of the stand-in")
