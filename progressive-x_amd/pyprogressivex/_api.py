"""Placeholder — replaced by the full drop-in API in the next milestone."""


def _todo(*a, **k):
    raise NotImplementedError("pyprogressivex API not wired yet")


find6DPoses = findHomographies = findTwoViewMotions = findFundamentalMatrices = findLines = findVanishingPoints = _todo
