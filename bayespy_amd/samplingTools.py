"""The one sampler the CLI reaches: drawCategory, used by the -M row cap
(DirichletEstimation/findDirichletPrior.py:85 -> DirichletEstimation/samplingTools.py:49-58).
Host-side by nature: it consumes Python's global `random` stream, one draw per decrement."""
import random
from itertools import accumulate


def drawCategory(distribution):
    """Index drawn with probability proportional to the (unnormalised) weights: the first index
    whose running sum exceeds one uniform draw scaled by the total, else the last index."""
    threshold = sum(distribution) * random.random()
    for index, running in enumerate(accumulate(distribution)):
        if threshold < running:
            return index
    return len(distribution) - 1
