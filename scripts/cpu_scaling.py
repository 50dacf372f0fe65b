"""How many host cores does this box really give us?  Oracle frame times vs thread count."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, chess2rt_amd as c2, oracle_lib as orc
print("usable_cores:", bench.usable_cores())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try: print(f, open(f).read().strip())
    except OSError as e: print(f, "absent")
s = c2.parseSceneFromFile(os.path.join(ROOT, "tests/golden/scenes/lecture5.sdl")); s.setFrameSize(1920, 1080); s.setAA(False)
cam = s.beginFrame(); o = s.renderOpts()
for th in (1, 4, 8, 16, 32, 64, 128, 256):
    st = {}
    orc.render_frame(s.desc, cam, o, th, st)
    t = time.time(); orc.render_frame(s.desc, cam, o, th, st); dt = time.time() - t
    print("%3d threads: %.3f s  %.1f Mray/s" % (th, dt, (st["primary"] + st["shadow"]) / dt / 1e6), flush=True)
