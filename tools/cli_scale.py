"""stan_solver on a generated n^3 STdb: phase times at a larger size than the tests use.
python tools/cli_scale.py [n] [fraction]: fraction > 0 removes that share of the elements (an irregular mesh:
stan_amd.cube.perforated_mesh; clamp x = 0, PointLoad on x = n), extra arguments go to stan_solver."""
import sys, os, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stan_amd import host
from stan_amd.cube import cube_mesh, cube_bcs, perforated_mesh
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
xyz, conn = perforated_mesh(n, frac) if frac > 0 else cube_mesh(n)
d = host.Db()
ne = conn.shape[0]
t0 = time.time()
d.set_mesh(np.arange(1, xyz.shape[0] + 1), xyz, np.arange(1, ne + 1), np.ones(ne), conn + 1, "HEX8_G2")
d.add_material(1, "Steel", 210000.0, 0.3); d.assign_part(1, 1, "HEX8_G2")
spc, ld, f = cube_bcs(n)
if frac > 0:
    spc = np.nonzero(xyz[:, 0] == 0.0)[0]
    ld = np.nonzero(xyz[:, 0] == float(n))[0]
d.add_bc(1, "fix", "SPC", spc + 1, np.ones((len(spc), 3)))
d.add_bc(2, "load", "PointLoad", ld + 1, np.tile(f, (len(ld), 1)))
d.set_analysis(tol=1e-6)
path = "/tmp/cli_scale.STdb"
d.write_stdb(path)
print("model %d^3 written: %.1f MB in %.1f s" % (n, os.path.getsize(path) / 1e6, time.time() - t0))
t0 = time.time()
out = subprocess.run([os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stan_amd", "bin", "stan_solver"), "--json"] + sys.argv[3:] + [path], capture_output=True, text=True)
print(out.stdout[-900:], out.stderr[-3000:])
print("stan_solver wall %.1f s, result file %.1f MB" % (time.time() - t0, os.path.getsize(path) / 1e6))
r = host.Db.read_stdb(path)
disp, e, s = r.results(1)
print("max |u| %.4e, max von-Mises-ish |s| %.3e" % (np.abs(disp).max(), np.abs(s).max()))
