"""The oracle's numpy-legacy RandomState clone against numpy itself and the G7 fixture
(recorded next to the reference run by tests/golden/make_golden.py)."""
import numpy
import pytest


@pytest.mark.parametrize("seed", [0, 1, 12345, 2**32 - 1])
def test_words_doubles_choice(oracle, golden, seed):
    fx = golden("g7_numpy_rng")
    r = oracle.Rng(seed)
    # numpy's randint(0, 2**32, dtype=uint32) takes one raw word each
    assert [r.u32() for _ in range(4)] == fx[f"seed{seed}_words"].tolist()
    r.seed(seed)
    got = numpy.array([r.double() for _ in range(700)])      # crosses the 624-word twist
    assert numpy.array_equal(got, fx[f"seed{seed}_doubles"])
    r.seed(seed)
    ks = (2, 3, 5, 7, 9, 4, 6, 8, 121, 1, 2)
    assert [r.below(k) for k in ks] == fx[f"seed{seed}_choice"].tolist()


@pytest.mark.parametrize("seed", [0, 1, 12345, 2**32 - 1])
@pytest.mark.parametrize("alpha,k", [(0.25, 2), (0.1, 9), (0.3, 7), (0.25, 4), (1.0, 3), (2.5, 5),
                                     (0.03, 121)])
def test_dirichlet_bit_exact(oracle, golden, seed, alpha, k):
    fx = golden("g7_numpy_rng")
    r = oracle.Rng(seed)
    got = numpy.array([r.dirichlet(alpha, k) for _ in range(6)])
    assert numpy.array_equal(got, fx[f"seed{seed}_dirichlet_{alpha}_{k}"])
    nxt = numpy.array([r.double(), r.double()])               # stream position afterwards
    assert numpy.array_equal(nxt, fx[f"seed{seed}_dirichlet_{alpha}_{k}_next"])


def test_survey_known_answers(oracle, golden):
    fx = golden("g7_numpy_rng")
    r = oracle.Rng(0)
    assert [r.below(2) for _ in range(10)] == [0, 1, 1, 0, 1, 1, 1, 1, 1, 1]
    assert fx["seed0_choice2x10"].tolist() == [0, 1, 1, 0, 1, 1, 1, 1, 1, 1]
    r.seed(0)
    assert numpy.array_equal(r.dirichlet(0.25, 2), fx["seed0_dirichlet_025x2"])
    # word consumption: 1-element choice 0 words, 2-element 1 word, choice(p=) 2 words
    r.seed(5)
    w0 = r.words
    r.below(1)
    assert r.words == w0
    r.below(2)
    assert r.words == w0 + 1
    r.choice_p([0.5, 0.5])
    assert r.words == w0 + 3


def test_choice_p(oracle, golden):
    fx = golden("g7_numpy_rng")
    r = oracle.Rng(3)
    picks = [10 + r.choice_p(p) for p in fx["seed3_choice_p"]]
    assert picks == fx["seed3_choice_p_picks"].tolist()


def test_live_against_numpy(oracle):
    """Direct comparison with the installed numpy (same library the reference calls)."""
    for seed in (7, 99, 2024):
        numpy.random.seed(seed)
        r = oracle.Rng(seed)
        for _ in range(300):
            k = int(numpy.random.randint(1, 12))
            assert r.below(12 - 1) + 1 == k            # randint(1,12) == 1 + below(11)
            a = numpy.random.dirichlet([0.3] * k)
            assert numpy.array_equal(a, r.dirichlet(0.3, k))
            assert numpy.random.random_sample() == r.double()
            assert numpy.random.choice(list(range(k))) == r.below(k)
        st = numpy.random.get_state()
        mine = r.get_numpy_state()
        assert numpy.array_equal(st[1], mine[1]) and st[2] == mine[2]


def test_state_roundtrip_with_numpy(oracle):
    numpy.random.seed(42)
    numpy.random.standard_normal(3)                     # leaves a cached gaussian
    st = numpy.random.get_state()
    r = oracle.Rng()
    r.set_numpy_state(st)
    assert numpy.array_equal(numpy.random.dirichlet([2.0] * 4), r.dirichlet(2.0, 4))
    numpy.random.set_state(r.get_numpy_state())
    assert numpy.random.random_sample() == r.double()
