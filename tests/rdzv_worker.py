"""One rank of tests/test_parallel_gloo.py::test_ranks_meet_at_the_launchers_store: fetches the launcher's key-value store the
way arvae_amd.parallel.connect() does before it builds the library's RCCL communicator, and passes 128 bytes from rank 0 to the
others through it (what arvae_comm_unique_id's bytes travel as).  No GPU, no process group."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arvae_amd import parallel  # noqa: E402

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
store = parallel._rendezvous_store(rank, world)
payload = bytes(range(128))
if rank == 0:
    store.set('arvae/comm/test', payload)
got = bytes(store.get('arvae/comm/test'))
assert got == payload
print(f'rank {rank} of {world} ok agent_store={os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "")}', flush=True)
