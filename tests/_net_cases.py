"""Topologies of the tests/golden/net_* fixtures, shared by the fixture generator (which passes
the reference's modules) and the parity tests (which pass this build's)."""


def _chain_graph(L, N_, N, r, f, C, out_at, legs):
    """A hand-built chain that is none of the layer classes: `legs` input nodes under every core
    (core k has one feature axis per leg) and the output label 'c' on core `out_at` (1-based)."""
    T = L.TensorNode
    mains, inputs = [], []
    for i in range(1, N + 1):
        shape, labels = [], []
        if i > 1:
            shape.append(r), labels.append(f"r{i}")
        if i == out_at:
            shape.append(C), labels.append("c")
        for leg in "pq"[:legs]:
            shape.append(f), labels.append(f"{leg}{i}")
        if i < N:
            shape.append(r), labels.append(f"r{i + 1}")
        a = T(tuple(shape), labels, l=f"r{i}" if i > 1 else None, r=f"r{i + 1}" if i < N else None, name=f"A{i}")
        for leg in "pq"[:legs]:
            x = T((1, f), ["s", f"{leg}{i}"], name=f"{leg.upper()}{i}")
            a.connect(x, f"{leg}{i}", priority=-1)
            inputs.append(x)
        mains.append(a)
    for a, b in zip(mains[:-1], mains[1:]):
        a.connect(b, a.right_labels[0], priority=1)
    return L.TensorNetworkLayer(N_.TensorNetwork(inputs, mains, output_labels=("s", "c")))


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
        # graphs outside every layer class: two input legs per core with the output on the middle core; a plain
        # chain with the output on its last core
        "graph_two_legs_c2": lambda: _chain_graph(L, N_, 3, 3, 6, 2, out_at=2, legs=2),
        "graph_out_last_c3": lambda: _chain_graph(L, N_, 3, 4, 6, 3, out_at=3, legs=1),
    }


NET_CASES = {"ttlinear_c2": ("xe", 2), "cumsum_c1": ("square", 1), "cpd_c1": ("square", 1), "cpd_c2": ("xe", 2),
             "sum_tt_type1_c1": ("square", 1), "sum_cpd_type1_c2": ("xe", 2), "sum_ttlinear_type1_c1": ("square", 1),
             "graph_two_legs_c2": ("xe", 2), "graph_out_last_c3": ("square", 3)}
