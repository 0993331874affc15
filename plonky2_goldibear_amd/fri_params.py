"""FriReductionStrategy::reduction_arity_bits (fri/reduction_strategies.rs:29-160): the list of FRI reduction arities a circuit's
FriParams carries.  The library derives the list of the stock strategy, ConstantArityBits, from the configuration by itself; a
circuit configured with Fixed(..) or MinSize(..) gets its list from here and hands it to gb_circuit_set_fri_reduction_arity_bits
(prover.CircuitData(reduction_arity_bits=...))."""


def constant_arity_bits(arity_bits, final_poly_bits, degree_bits, rate_bits, cap_height):
    """ConstantArityBits(arity_bits, final_poly_bits) (:44-56): reduce by 2^arity_bits until the degree is at most 2^final_poly_bits
    or one more reduction would make the last FRI tree lower than cap_height."""
    out = []
    while degree_bits > final_poly_bits:
        # usize arithmetic there: the subtraction panics in a debug build and wraps in a release build, where the assert below
        # then fails - the reference does not survive this configuration either way
        assert degree_bits + rate_bits >= arity_bits, "attempt to subtract with overflow (fri/reduction_strategies.rs:42)"
        if degree_bits + rate_bits - arity_bits < cap_height:
            break
        out.append(arity_bits)
        assert degree_bits >= arity_bits, "assertion failed: degree_bits >= arity_bits (fri/reduction_strategies.rs:45)"
        degree_bits -= arity_bits
    return out


def relative_proof_size(degree_bits, rate_bits, num_queries, arity_bits):
    """:129-160 - approximate FRI proof size in field elements (D = 4 there, whatever the field's extension degree)"""
    D = 4
    current_layer_bits = degree_bits + rate_bits
    total = 0
    for ab in arity_bits:
        total += ((1 << ab) - 1) * D * num_queries       # neighbouring evaluations
        total += current_layer_bits * 4 * num_queries    # siblings in the Merkle path
        current_layer_bits -= ab
    assert current_layer_bits >= rate_bits
    return total + D * (1 << (current_layer_bits - rate_bits))


def _min_size_helper(degree_bits, rate_bits, num_queries, global_max_arity_bits, prefix):
    """:85-127: depth-first over monotonically non-increasing arity sequences; first strictly smaller size wins"""
    current_layer_bits = degree_bits + rate_bits - sum(prefix)
    assert current_layer_bits >= rate_bits
    best, best_size = list(prefix), relative_proof_size(degree_bits, rate_bits, num_queries, prefix)
    max_ab = min(prefix[-1] if prefix else global_max_arity_bits, current_layer_bits - rate_bits)
    for nxt in range(1, max_ab + 1):
        cand, size = _min_size_helper(degree_bits, rate_bits, num_queries, max_ab, prefix + [nxt])
        if size < best_size:
            best, best_size = cand, size
    return best, best_size


def min_size_arity_bits(degree_bits, rate_bits, num_queries, max_arity_bits=None):
    """MinSize(opt_max_arity_bits) (:58-83); None = the reference's default of 4"""
    return _min_size_helper(degree_bits, rate_bits, num_queries, 4 if max_arity_bits is None else max_arity_bits, [])[0]


def reduction_arity_bits(strategy, degree_bits, rate_bits, cap_height, num_queries):
    """strategy: ("constant", arity_bits, final_poly_bits) | ("fixed", [bits..]) | ("min_size", max_arity_bits or None)"""
    kind = strategy[0]
    if kind == "fixed":
        return list(strategy[1])
    if kind == "constant":
        return constant_arity_bits(strategy[1], strategy[2], degree_bits, rate_bits, cap_height)
    if kind == "min_size":
        return min_size_arity_bits(degree_bits, rate_bits, num_queries, strategy[1] if len(strategy) > 1 else None)
    raise ValueError("unknown FriReductionStrategy %r" % (kind,))
