import time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import oracle as O
from oracle import plonk_dummy as D
print("threads", O.lib().gbo_num_threads(), flush=True)
circ = D.DummyCircuit(16, D.CircuitConfig(num_challenges=3), check_security=False)
_ = circ.circuit_digest
w = circ.witness(seed=1)
t2 = time.perf_counter(); proof, dbg = D.prove_cpu(circ, w); t3 = time.perf_counter()
print("2^16 prove %.2f cs %.2f" % (t3 - t2, D.prove_cpu.last_cs_commit_seconds), flush=True)
