"""Host-side logic of the drop-in surface (topology planning, environment caching, sweep
schedule, conv columns) run on the CPU with the kernel stand-ins of tests/_kernel_standins.py
against the reference's own outputs.  The kernels themselves are covered by the GPU tests."""
import pytest

from tests import _golden as G
from tests import _kernel_standins
from tests import test_conv_gpu as conv


@pytest.fixture
def cpu_kernels(monkeypatch):
    _kernel_standins.install(monkeypatch)


@pytest.mark.parametrize("name", G.names("conv_tt_"))
def test_conv_host_logic(name, cpu_kernels):
    conv.check_local_problems(name, "cpu")
    conv.check_cholesky_sweep(name, "cpu")
    conv.check_lanczos_sweep(name, "cpu")


@pytest.mark.parametrize("name", G.names("convfull_"))
def test_conv_full_shape_host_logic(name, cpu_kernels):
    conv.check_full_shape(name, "cpu")


@pytest.fixture
def api(monkeypatch, cpu_kernels):
    from tests import test_api_gpu
    monkeypatch.setattr(test_api_gpu, "DEV", "cpu")
    return test_api_gpu


@pytest.mark.parametrize("name", G.names("tt_core_update_"))
def test_core_update_host_logic(name, api):
    api.test_core_update_golden(name)


@pytest.mark.parametrize("name", G.names("swipe_trace_"))
def test_swipe_schedule_host_logic(name, api):
    api.test_swipe_trace_golden(name)


@pytest.mark.parametrize("name", G.names("lanczos_") + G.names("scipy_"))
def test_krylov_sweeps_host_logic(name, api):
    (api.test_lanczos_golden if name.startswith("lanczos_") else api.test_scipy_golden)(name)


@pytest.mark.parametrize("name", G.names("tnml_fit_") + G.names("ttreg_fit_"))
def test_estimators_host_logic(name, api):
    (api.test_tnml_fit_golden if name.startswith("tnml_") else api.test_ttreg_fit_golden)(name)


@pytest.mark.parametrize("case", range(16))
def test_random_conv_host_logic(case, cpu_kernels):
    conv.check_random_conv(case, "cpu")


@pytest.mark.parametrize("case", range(24))
def test_random_swipes_host_logic(case, api):
    api.test_random_swipe_golden(case)


@pytest.mark.parametrize("case", range(24))
def test_random_estimators_host_logic(case, api):
    api.test_random_estimator_golden(case)


@pytest.mark.parametrize("name", G.names("net_"))
def test_other_topologies_host_logic(name, cpu_kernels):
    from tests import test_nets_gpu as nets
    nets.check_local_problems(name, "cpu")
    nets.check_cholesky_sweep(name, "cpu")
    nets.check_lanczos_sweep(name, "cpu")


def test_conv_many_patches_host_logic(cpu_kernels, monkeypatch):
    conv.check_many_patches("cpu", monkeypatch)


def _meddling_sweep(device, speculate, poke_gram=False):
    """Two sweeps during which a callback changes nodes behind the engine's back: the node visited
    next is scaled in place after every update, and one node is replaced by a new tensor."""
    import torch
    from tensornetworksfork_amd.tensor.layers import TensorTrainLayer
    from tensornetworksfork_amd.tensor.bregman import SquareBregFunction
    from tensornetworksfork_amd.tensor.network import TensorNetwork
    torch.set_default_dtype(torch.float64)
    layer = TensorTrainLayer(5, 3, 3, output_shape=1, seed=3).to(device)
    tn = layer.tensor_network
    used = []
    if not speculate:
        tn._speculate = lambda *a, **k: None
    else:
        orig = TensorNetwork._site_front

        def counting_front(self, *a, **k):
            used.append(1)
            return orig(self, *a, **k)
        tn._site_front = counting_front.__get__(tn)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(60, 3, generator=g).to(device)
    y = torch.rand(60, 1, generator=g).to(device)
    visits = []

    def meddle(NS, node):
        visits.append(node)
        i = tn.main_nodes.index(node)
        nxt = tn.main_nodes[(i + 1) % len(tn.main_nodes)]
        nxt.tensor.mul_(1.01)                                   # in place: same tensor object, new version
        if len(visits) == 3:
            tn.main_nodes[0].tensor = tn.main_nodes[0].tensor * 0.99   # a new tensor object
    def poke(NS, node):
        i = tn.main_nodes.index(node)
        other = tn.main_nodes[1 if i != 1 else 2]      # nodes 1..3 share one shape
        yp = tn.forward(x, to_tensor=True)
        _, gg, HH = SquareBregFunction().forward(yp, y)
        tn.get_A_b(other, gg, HH)

    assert tn.accumulating_swipe(x, y, SquareBregFunction(), method="ridge_cholesky", eps=1.0, num_swipes=2,
                                 batch_size=16, block_callback=(meddle if not poke_gram else poke))
    return [n.tensor.detach().cpu().clone() for n in tn.main_nodes], len(used)


def test_speculative_front_is_dropped_when_nodes_change(cpu_kernels):
    import torch
    ref, _ = _meddling_sweep("cpu", speculate=False)
    got, fronts = _meddling_sweep("cpu", speculate=True)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    assert fronts > len(ref)      # every queued front was dropped and recomputed


def test_gradient_swipe_takes_the_cifar_call(cpu_kernels):
    conv.check_gradient_swipe_cifar_call("cpu")


def test_pair_classes_follow_the_tensor_not_its_address(cpu_kernels):
    """The class cache must not hand the classes of one feature tensor to another tensor, whatever
    address, shape or version the two share; views of one tensor share an entry."""
    import torch
    from tensornetworksfork_amd.tensor.layers import TensorTrainLayer
    torch.set_default_dtype(torch.float64)
    eng = TensorTrainLayer(3, 2, 3, output_shape=1, seed=1).tensor_network._engine
    x = torch.rand(300, 1) * 2 - 1
    poly = torch.cat([x ** k for k in range(4)], dim=1).contiguous()       # classes by p + q
    rep = eng.pair_rep(poly.reshape(1, 1, 300, 4)[0, 0])
    assert rep is not None and len(set(rep)) == 7
    assert eng.pair_rep(poly.reshape(1, 1, 300, 4)[0, 0]) is rep            # a fresh view of the same owner hits
    # another tensor with identical metadata (what a recycled allocation looks like) must miss
    other = torch.rand(300, 4)
    assert other._version == poly._version and other.shape == poly.shape
    assert eng.pair_rep(other) is None
    poly.mul_(1.0)                                                           # new version: recomputed, same classes
    assert eng.pair_rep(poly) == rep
    eng.drop_pairs([poly])
    assert all(e[0] is not poly for e in eng._pairs.values())


@pytest.mark.gpu
def test_speculative_front_survives_a_gram_from_a_callback():
    """A callback that computes the Gram matrix of another node of the same shape between two
    updates overwrites the shared workspace without changing any parameter: the queued front must
    be dropped, the sweep must give what it gives without speculation."""
    import torch
    ref, _ = _meddling_sweep("cuda:0", speculate=False, poke_gram=True)
    got, _ = _meddling_sweep("cuda:0", speculate=True, poke_gram=True)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
