"""Kuhn-Munkres assignment, restated (oracle, test infrastructure).

The reference gets this from the third-party PyPI package ``munkres``
(``from munkres import Munkres``; ``Munkres().compute(cost)`` at
rtpe/third_party/group.py:14,19-23).  The package is not vendored in
/root/reference, carries no version pin there (README.md:10 only says "same as
HigherHRNet"), and is not installed in this image.  Its *published procedure*
(the 1.1.x line, current when the reference was written) is restated here,
including the scan orders that decide equal-cost ties - and ties are common on
this path, because group.py:66 builds costs as ``round(dist)*100 - val``, so
swapping two columns changes the total by multiples of 100 that can cancel
exactly:

  0. pad to a square matrix with zeros;
  1. subtract each row's minimum;
  2. star, row by row, the first zero whose row and column hold no star yet;
  3. cover every column that holds a star; all n covered -> finished;
  4. repeatedly look for an uncovered zero, scanning rows cyclically from the
     row of the previous find and, inside a row, columns cyclically from the
     previous column, taking the LAST uncovered zero of the first row that has
     one (the package's inner loop does not stop at the first hit); prime it;
     if its row holds a star, cover the row and uncover the star's column,
     else go to 5; no uncovered zero left -> 6;
  5. alternate primed / starred zeros from the uncovered prime, flip them,
     erase primes, clear covers, back to 3;
  6. m = smallest uncovered value; add m to every covered row, subtract it
     from every uncovered column (two separate float operations for an element
     in both), back to 4 with the scan restarted at (0, 0).

Results are the starred cells inside the original shape, in row-major order.
"parity unpinned": this cannot be run against the real package here; for
unique optima any solver agrees (tests check optimality against scipy and
brute force), for ties the answer follows the orders above.

``rtpe_match_by_tag`` in csrc/match_host.cpp implements the same procedure in
C++ and is compared against this file in tests/.
"""
import itertools

import numpy as np


def munkres_compute(cost):
    """cost: 2-D array-like (rows x cols, float).  Returns list of (r, c)."""
    cost = np.asarray(cost, dtype=np.float64)
    if cost.size == 0:
        return []
    nr, nc = cost.shape
    n = max(nr, nc)
    C = [[0.0] * n for _ in range(n)]
    for i in range(nr):
        for j in range(nc):
            C[i][j] = float(cost[i, j])
    row_cov = [False] * n
    col_cov = [False] * n
    mark = [[0] * n for _ in range(n)]        # 1 = starred, 2 = primed

    for i in range(n):                        # step 1
        m = min(C[i])
        for j in range(n):
            C[i][j] -= m
    for i in range(n):                        # step 2
        for j in range(n):
            if C[i][j] == 0 and not col_cov[j] and not row_cov[i]:
                mark[i][j] = 1
                col_cov[j] = True
                row_cov[i] = True
                break
    row_cov = [False] * n
    col_cov = [False] * n

    def find_zero(i0, j0):
        i = i0
        while True:
            hit = -1
            j = j0
            while True:
                if C[i][j] == 0 and not row_cov[i] and not col_cov[j]:
                    hit = j                    # keeps the last one of this row
                j = (j + 1) % n
                if j == j0:
                    break
            if hit >= 0:
                return i, hit
            i = (i + 1) % n
            if i == i0:
                return -1, -1

    step = 3
    while True:
        if step == 3:
            count = 0
            for i in range(n):
                for j in range(n):
                    if mark[i][j] == 1 and not col_cov[j]:
                        col_cov[j] = True
                        count += 1
            if count >= n:
                break
            step = 4
        elif step == 4:
            row, col = 0, 0
            while True:
                row, col = find_zero(row, col)
                if row < 0:
                    step = 6
                    break
                mark[row][col] = 2
                star = next((j for j in range(n) if mark[row][j] == 1), -1)
                if star >= 0:
                    col = star
                    row_cov[row] = True
                    col_cov[col] = False
                else:
                    z0 = (row, col)
                    step = 5
                    break
        elif step == 5:
            path = [z0]
            while True:
                r = next((i for i in range(n) if mark[i][path[-1][1]] == 1), -1)
                if r < 0:
                    break
                path.append((r, path[-1][1]))
                c = next(j for j in range(n) if mark[r][j] == 2)
                path.append((r, c))
            for (r, c) in path:
                mark[r][c] = 0 if mark[r][c] == 1 else 1
            row_cov = [False] * n
            col_cov = [False] * n
            for i in range(n):
                for j in range(n):
                    if mark[i][j] == 2:
                        mark[i][j] = 0
            step = 3
        else:                                  # step 6
            m = min(C[i][j] for i in range(n) for j in range(n)
                    if not row_cov[i] and not col_cov[j])
            for i in range(n):
                for j in range(n):
                    if row_cov[i]:
                        C[i][j] += m
                    if not col_cov[j]:
                        C[i][j] -= m
            step = 4
    return [(i, j) for i in range(nr) for j in range(nc) if mark[i][j] == 1]


def brute_force_cost(cost):
    """minimum total cost over all assignments (small sizes) - test helper"""
    cost = np.asarray(cost, np.float64)
    nr, nc = cost.shape
    best = np.inf
    if nr <= nc:
        for cols in itertools.permutations(range(nc), nr):
            best = min(best, sum(cost[i, c] for i, c in enumerate(cols)))
    else:
        for rows in itertools.permutations(range(nr), nc):
            best = min(best, sum(cost[r, j] for j, r in enumerate(rows)))
    return best
