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


@pytest.mark.parametrize("name", G.names("net_"))
def test_other_topologies_host_logic(name, cpu_kernels):
    from tests import test_nets_gpu as nets
    nets.check_local_problems(name, "cpu")
    nets.check_cholesky_sweep(name, "cpu")
    nets.check_lanczos_sweep(name, "cpu")


def test_conv_many_patches_host_logic(cpu_kernels, monkeypatch):
    conv.check_many_patches("cpu", monkeypatch)
