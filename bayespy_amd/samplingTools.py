"""The one sampler the CLI reaches: drawCategory, used by the -M row cap
(DirichletEstimation/findDirichletPrior.py:85, samplingTools.py:49-58).  Host-side by nature
(it consumes Python's global `random` stream, one draw per decrement)."""
import random


# Draws a category from an unnormalized distribution
def drawCategory(distribution):
    K = len(distribution)
    r = sum(distribution) * random.random()
    runningTotal = 0
    for k in range(0, K):
        runningTotal += distribution[k]
        if (r < runningTotal):
            return k
    return K - 1
