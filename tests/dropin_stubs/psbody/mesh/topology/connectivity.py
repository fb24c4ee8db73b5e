"""Stand-in for psbody.mesh.topology.connectivity (imported by the reference's lib/mesh_sampling.py:120,244): the two
connectivity helpers, answered by cape_amd.mesh_operators.  TEST INFRASTRUCTURE ONLY."""
from cape_amd.mesh_operators import get_vert_connectivity, get_vertices_per_edge   # noqa: F401
