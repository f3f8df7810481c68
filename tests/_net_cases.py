"""Topologies of the tests/golden/net_* fixtures, shared by the fixture generator (which passes
the reference's modules) and the parity tests (which pass this build's)."""


def net_builders(L, N_):
    """case name -> layer factory, over a module `L` with the reference's layer constructors and
    `N_` with its network classes (the tests call this with the build's own modules)."""
    def type1(make):
        def build():
            nets = [make(i).tensor_network for i in range(1, 4)]
            return L.TensorNetworkLayer(N_.SumOfNetworks(nets, output_labels=nets[0].output_labels, train_operators=True))
        return build
    return {
        "ttlinear_c2": lambda: L.TensorTrainLinearLayer(3, 4, 6, 3, output_shape=2, constrict_bond=False, seed=51),
        "cumsum_c1": lambda: L.CumSumLayer(3, 4, 6, output_shape=1, constrict_bond=False),
        "cpd_c1": lambda: L.CPDLayer(3, 4, 6, output_shape=1, seed=53),
        "cpd_c2": lambda: L.CPDLayer(3, 4, 6, output_shape=2, seed=54),
        "sum_tt_type1_c1": type1(lambda i: L.TensorTrainLayer(i, 4, 5 if i != 1 else 6, output_shape=1, constrict_bond=False,
                                                              seed=60 + i)),
        "sum_cpd_type1_c2": type1(lambda i: L.CPDLayer(i, 3, 5 if i != 1 else 6, output_shape=2, seed=70 + i)),
        "sum_ttlinear_type1_c1": type1(lambda i: L.TensorTrainLinearLayer(i, 4, 5 if i != 1 else 6, 3, output_shape=1,
                                                                        constrict_bond=False, seed=80 + i)),
    }


NET_CASES = {"ttlinear_c2": ("xe", 2), "cumsum_c1": ("square", 1), "cpd_c1": ("square", 1), "cpd_c2": ("xe", 2),
             "sum_tt_type1_c1": ("square", 1), "sum_cpd_type1_c2": ("xe", 2), "sum_ttlinear_type1_c1": ("square", 1)}
