"""sorted-merge tie rule (dru.clj:82-104): the O(N log U) form used by the oracle at scale must reproduce the
literal stable-sort + cons-to-front restatement on tie-heavy inputs."""
import numpy as np


def test_worked_examples(oracle):
    # A:[1,2], B:[1,2] -> A1 B1 B2 A2 ;  A:[1,1], B:[1,1] -> A1 A2 B1 B2
    for lit in (True, False):
        assert list(oracle.sorted_merge([[1.0, 2.0], [1.0, 2.0]], literal=lit)) == [0, 1, 1, 0]
        assert list(oracle.sorted_merge([[1.0, 1.0], [1.0, 1.0]], literal=lit)) == [0, 0, 1, 1]
        assert list(oracle.sorted_merge([[1.0, 2.0, 3.0], [1.0, 2.0, 3.0]], literal=lit)) == [0, 1, 1, 0, 0, 1]
        assert list(oracle.sorted_merge([[2.0], [1.0, 2.0], [0.5, 2.0]], literal=lit)) == [2, 1, 1, 2, 0]


def test_heap_equals_literal_random(oracle):
    rng = np.random.default_rng(7)
    for trial in range(300):
        U = int(rng.integers(1, 12))
        colls = []
        for _ in range(U):
            n = int(rng.integers(0, 9))
            steps = rng.integers(0 if trial % 3 == 0 else 1, 3, size=n)  # small integer steps -> many ties
            colls.append(list(np.cumsum(steps).astype(float)))
        a = oracle.sorted_merge(colls, literal=True)
        b = oracle.sorted_merge(colls, literal=False)
        assert np.array_equal(a, b), (colls, a, b)
