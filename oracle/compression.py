"""TEST ORACLE - compressed proofs (ProofWithPublicInputs::compress / CompressedProofWithPublicInputs::decompress).

Test infrastructure only.  Follows (paths relative to /root/reference/plonky2/src):
  hash/path_compression.rs:12-117     compress_merkle_proofs / decompress_merkle_proofs
  fri/proof.rs:137-232, 237-384       FriProof::compress / CompressedFriProof::decompress
  plonk/proof.rs:96-140, 183-260      ProofWithPublicInputs::compress, CompressedProofWithPublicInputs::{decompress, verify}
  plonk/get_challenges.rs:103-180     fri_query_indices / get_inferred_elements
  util/serialization/mod.rs:1102-1150, 2168-2256   byte layout of the compressed form
The reference holds no serialized compressed proof; what pins this file is the round trip on the reference's regression proof
(tests/test_compression.py): compress -> decompress reproduces its bytes exactly (149044 -> 137620 -> 149044).
"""
import struct

from . import verifier as V
from .fields import GL


# ----------------------------------------------------------------------------- hash/path_compression.rs
def compress_merkle_proofs(cap_height, indices, proofs):
    height = cap_height + len(proofs[0])
    num_leaves = 1 << height
    known = set()
    for i in indices:
        for j in range(height - cap_height):
            known.add((i + num_leaves) >> j)
    out = []
    for i, p in zip(indices, proofs):
        comp, index = [], i + num_leaves
        for sibling in p:
            if (index ^ 1) not in known:
                comp.append(sibling)
                known.add(index ^ 1)
            index >>= 1
            known.add(index)
        out.append(comp)
    return out


def decompress_merkle_proofs(F, leaves_data, leaves_indices, compressed_proofs, height, cap_height):
    num_leaves = 1 << height
    seen = {}
    for i, v in zip(leaves_indices, leaves_data):
        seen[i + num_leaves] = [int(x) for x in F.mod.hash_or_noop(_arr(F, v))]
    its = [iter(p) for p in compressed_proofs]
    for layer in range(height - cap_height):
        for i, it in zip(leaves_indices, its):
            index = (i + num_leaves) >> layer
            cur = seen[index]
            sib_index = index ^ 1
            if sib_index not in seen:
                seen[sib_index] = next(it)
            sib = seen[sib_index]
            l, r = (cur, sib) if index % 2 == 0 else (sib, cur)
            seen[index >> 1] = [int(x) for x in F.mod.two_to_one(_arr(F, l), _arr(F, r))]
    # (a repeated index carries its first occurrence's siblings a second time; they stay unread, as in the reference)
    out = []
    for i in leaves_indices:
        index, path = i + num_leaves, []
        for _ in range(height - cap_height):
            path.append(seen[index ^ 1])
            index >>= 1
        out.append(path)
    return out


def _arr(F, v):
    import numpy as np
    return np.asarray([int(x) for x in v], dtype=F.dtype)


# ----------------------------------------------------------------------------- fri/proof.rs compress
def compress_fri_proof(fri, indices, cd):
    """-> dict(commit_phase_merkle_caps, indices, initial_trees_proofs {index: [(vals, path)]}, steps [{index: (evals, path)}],
    final_poly, pow_witness)"""
    fp = cd["fri_params"]
    cap_height = cd["config"]["fri_config"]["cap_height"]
    arity = fp["reduction_arity_bits"]
    qrps = fri["query_round_proofs"]
    ntrees = len(qrps[0]["initial_trees_proof"])
    it_idx = [[] for _ in range(ntrees)]
    it_leaves = [[] for _ in range(ntrees)]
    it_proofs = [[] for _ in range(ntrees)]
    st_idx = [[] for _ in arity]
    st_evals = [[] for _ in arity]
    st_proofs = [[] for _ in arity]
    for index, q in zip(indices, qrps):
        for t, (vals, path) in enumerate(q["initial_trees_proof"]):
            it_idx[t].append(index)
            it_leaves[t].append(vals)
            it_proofs[t].append(path)
        for i, (evals, path) in enumerate(q["steps"]):
            within = index & ((1 << arity[i]) - 1)
            index >>= arity[i]
            st_idx[i].append(index)
            st_evals[i].append(evals[:within] + evals[within + 1:])   # the element the verifier can infer is removed
            st_proofs[i].append(path)
    it_proofs = [compress_merkle_proofs(cap_height, is_, ps) for is_, ps in zip(it_idx, it_proofs)]
    st_proofs = [compress_merkle_proofs(cap_height, is_, ps) for is_, ps in zip(st_idx, st_proofs)]
    out = dict(commit_phase_merkle_caps=fri["commit_phase_merkle_caps"], indices=list(indices), initial_trees_proofs={},
               steps=[{} for _ in arity], final_poly=fri["final_poly"], pow_witness=fri["pow_witness"])
    for i, index in enumerate(indices):
        out["initial_trees_proofs"].setdefault(index, [(it_leaves[t][i], it_proofs[t][i]) for t in range(ntrees)])
        for j in range(len(arity)):
            index >>= arity[j]
            out["steps"][j].setdefault(index, (st_evals[j][i], st_proofs[j][i]))
    return out


def compress(pr, pis, circuit_digest, cd, F=GL):
    """ProofWithPublicInputs::compress (plonk/proof.rs:96-118): the query indices come from the transcript"""
    ch = V.get_challenges(pr, pis, circuit_digest, cd, F)
    out = {k: pr[k] for k in ("wires_cap", "zs_cap", "quotient_cap", "openings")}
    out["opening_proof"] = compress_fri_proof(pr["opening_proof"], ch["fri_query_indices"], cd)
    return out


# ----------------------------------------------------------------------------- byte layout
def write_compressed_proof_with_pis(cpr, pis, F=GL):
    out = bytearray()
    fmt = "<Q" if F.elem_bytes == 8 else "<I"

    def fv(xs):
        for x in xs:
            out.extend(struct.pack(fmt, int(x)))

    def ev(xs):
        for x in xs:
            fv(x)

    def cap(c):
        for h in c:
            fv(h)

    def mp(p):
        out.append(len(p))
        for h in p:
            fv(h)

    cap(cpr["wires_cap"]); cap(cpr["zs_cap"]); cap(cpr["quotient_cap"])
    o = cpr["openings"]
    for k in ("constants", "plonk_sigmas", "wires", "plonk_zs", "plonk_zs_next", "lookup_zs", "lookup_zs_next",
              "partial_products", "quotient_polys"):
        ev(o[k])
    fri = cpr["opening_proof"]
    for c in fri["commit_phase_merkle_caps"]:
        cap(c)
    for i in fri["indices"]:
        out.extend(struct.pack("<I", i))
    for _, itp in sorted(fri["initial_trees_proofs"].items()):
        for vals, path in itp:
            fv(vals); mp(path)
    for h in fri["steps"]:
        for _, (evals, path) in sorted(h.items()):
            ev(evals); mp(path)
    ev(fri["final_poly"])
    fv([fri["pow_witness"]])
    fv(pis)   # no length prefix: the reader takes what remains (serialization/mod.rs:2255, 1230-1236)
    return bytes(out)


def read_compressed_proof_with_pis(data, cd, F=GL):
    r = V.Reader(data, F)
    cfg, fp = cd["config"], cd["fri_params"]
    ch = cfg["fri_config"]["cap_height"]
    c = cfg["num_challenges"]
    salt = V.SALT_SIZE if fp["hiding"] else 0
    nlk = c * cd["num_lookup_polys"]
    pr = dict(wires_cap=r.cap(ch), zs_cap=r.cap(ch), quotient_cap=r.cap(ch))
    pr["openings"] = dict(
        constants=r.ext_vec(cd["num_constants"]), plonk_sigmas=r.ext_vec(cfg["num_routed_wires"]),
        wires=r.ext_vec(cfg["num_wires"]), plonk_zs=r.ext_vec(c), plonk_zs_next=r.ext_vec(c),
        lookup_zs=r.ext_vec(nlk), lookup_zs_next=r.ext_vec(nlk),
        partial_products=r.ext_vec(cd["num_partial_products"] * c),
        quotient_polys=r.ext_vec(cd["quotient_degree_factor"] * c))
    fri = dict(commit_phase_merkle_caps=[r.cap(ch) for _ in fp["reduction_arity_bits"]])
    widths = [cd["num_constants"] + cfg["num_routed_wires"], cfg["num_wires"] + salt,
              c * (1 + cd["num_partial_products"] + cd["num_lookup_polys"]) + salt, c * cd["quotient_degree_factor"] + salt]
    original = [r.u32() for _ in range(cfg["fri_config"]["num_query_rounds"])]
    indices = sorted(set(original))
    fri["indices"] = original
    fri["initial_trees_proofs"] = {i: [(r.field_vec(w), r.merkle_proof()) for w in widths] for i in indices}
    fri["steps"] = []
    for ab in fp["reduction_arity_bits"]:
        indices = sorted({x >> ab for x in indices})
        fri["steps"].append({i: (r.ext_vec((1 << ab) - 1), r.merkle_proof()) for i in indices})
    fri["final_poly"] = r.ext_vec(1 << (fp["degree_bits"] - sum(fp["reduction_arity_bits"])))
    fri["pow_witness"] = r.field()
    pr["opening_proof"] = fri
    pis = r.field_vec((len(data) - r.o) // F.elem_bytes)
    assert r.done(), "trailing bytes in compressed proof"
    return pr, pis


# ----------------------------------------------------------------------------- decompress
def decompress(cpr, pis, circuit_digest, cd, F=GL):
    """CompressedProofWithPublicInputs::decompress (plonk/proof.rs:183-215): challenges from the transcript (the compressed
    proof carries its indices, which must be the transcript's), the inferred elements (get_challenges.rs:129-180: the value
    every FRI layer's fold must reproduce), then CompressedFriProof::decompress."""
    cfri = cpr["opening_proof"]
    shell = dict(cpr)
    shell["opening_proof"] = dict(commit_phase_merkle_caps=cfri["commit_phase_merkle_caps"], final_poly=cfri["final_poly"],
                                  pow_witness=cfri["pow_witness"])
    ch = V.get_challenges(shell, pis, circuit_digest, cd, F)
    indices = ch["fri_query_indices"]
    assert indices == cfri["indices"], "the compressed proof's query indices are not the transcript's"
    cfg, fp = cd["config"], cd["fri_params"]
    cap_height = cfg["fri_config"]["cap_height"]
    arity = fp["reduction_arity_bits"]
    height = fp["degree_bits"] + cfg["fri_config"]["rate_bits"]
    P_ = F.P
    blinding, batches = V.fri_instance(cd, ch["plonk_zeta"], F)
    alpha = ch["fri_alpha"]
    reduced_openings = [V.reduce_with_alpha(alpha, b, F)[0] for b in V.fri_openings(cpr["openings"])]
    ntrees = len(next(iter(cfri["initial_trees_proofs"].values())))
    it_idx = [[] for _ in range(ntrees)]
    it_leaves = [[] for _ in range(ntrees)]
    it_proofs = [[] for _ in range(ntrees)]
    st_idx = [[] for _ in arity]
    st_evals = [[] for _ in arity]
    st_proofs = [[] for _ in arity]
    evals_by_depth = [{} for _ in arity]
    for x_index in indices:
        itp = cfri["initial_trees_proofs"][x_index]
        for t, (vals, path) in enumerate(itp):
            it_idx[t].append(x_index)
            it_leaves[t].append(vals)
            it_proofs[t].append(path)
        # fri_combine_initial (fri/verifier.rs:121-165)
        subgroup_x = F.generator * pow(F.two_adic_generator(height), V.reverse_bits(x_index, height), P_) % P_
        total = F.zero
        for (point, polys), red_open in zip(batches, reduced_openings):
            evs = []
            for (oi, pi) in polys:
                vals = itp[oi][0]
                salted = fp["hiding"] and blinding[oi]
                evs.append(F.efrom(vals[: len(vals) - (V.SALT_SIZE if salted else 0)][pi]))
            red, count = V.reduce_with_alpha(alpha, evs, F)
            total = F.emul(F.epow(alpha, count), total)
            total = F.eadd(total, F.ediv(F.esub(red, red_open), F.esub(F.efrom(subgroup_x), point)))
        old_eval, index = total, x_index
        for i, ab in enumerate(arity):
            within = index & ((1 << ab) - 1)
            index >>= ab
            evals, path = cfri["steps"][i][index]
            st_idx[i].append(index)
            if index in evals_by_depth[i]:
                evals = evals_by_depth[i][index]
            else:
                evals = list(evals[:within]) + [old_eval] + list(evals[within:])   # the inferred element
                evals_by_depth[i][index] = evals
            old_eval = V.compute_evaluation(subgroup_x, within, ab, evals, ch["fri_betas"][i], F)
            subgroup_x = pow(subgroup_x, 1 << ab, P_)
            st_evals[i].append(evals)
            st_proofs[i].append(path)
    it_paths = [decompress_merkle_proofs(F, ls, is_, ps, height, cap_height) for ls, is_, ps in zip(it_leaves, it_idx, it_proofs)]
    heights, h = [], height
    for ab in arity:
        h -= ab
        heights.append(h)
    st_paths = [decompress_merkle_proofs(F, [[x for e in ev for x in e] for ev in evs], is_, ps, hh, cap_height)
                for evs, is_, ps, hh in zip(st_evals, st_idx, st_proofs, heights)]
    qrps = []
    for i in range(len(indices)):
        qrps.append(dict(initial_trees_proof=[(it_leaves[t][i], it_paths[t][i]) for t in range(ntrees)],
                         steps=[(st_evals[j][i], st_paths[j][i]) for j in range(len(arity))]))
    out = {k: cpr[k] for k in ("wires_cap", "zs_cap", "quotient_cap", "openings")}
    out["opening_proof"] = dict(commit_phase_merkle_caps=cfri["commit_phase_merkle_caps"], query_round_proofs=qrps,
                                final_poly=cfri["final_poly"], pow_witness=cfri["pow_witness"])
    return out


def compress_bytes(proof_bytes, circuit_digest, cd, F=GL):
    pr, pis = V.read_proof_with_pis(proof_bytes, cd, F)
    return write_compressed_proof_with_pis(compress(pr, pis, circuit_digest, cd, F), pis, F)


def decompress_bytes(compressed_bytes, circuit_digest, cd, F=GL):
    cpr, pis = read_compressed_proof_with_pis(compressed_bytes, cd, F)
    return V.write_proof_with_pis(decompress(cpr, pis, circuit_digest, cd, F), pis, F)
