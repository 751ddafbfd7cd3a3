"""CPU: the resident server of mhx_core (`--serve <socket>`, clients with MHX_SERVER=<socket>): request framing, the
client's stderr / working directory / MHX_* environment reaching the server, status propagation, stop, autostart and
the fall-back to doing the work in the client when no server listens.  Exercised through `buildlib` with the sequential
host parser (MHX_BUILDLIB_HOST=1: no GPU involved); the GPU sub-programs go through the same path in
tests/test_gpu_server.py."""
import os
import subprocess
import time

import pytest

import golden_util as gu
from megahit_amd import canon
from test_front_process import write_inputs

pytestmark = pytest.mark.skipif(not os.path.exists(gu.MHX_CORE), reason="needs mhx_core")


def run(args, env, cwd=None):
    e = dict(os.environ)
    e.update(env)
    return subprocess.run([gu.MHX_CORE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e, timeout=60, cwd=cwd)


@pytest.fixture
def server(tmp_path):
    sock = str(tmp_path / "mhx.sock")
    log = open(str(tmp_path / "server.log"), "w")
    p = subprocess.Popen([gu.MHX_CORE, "--serve", sock], stdout=subprocess.DEVNULL, stderr=log, env=dict(os.environ, MHX_SERVE_IDLE_S="30"))
    for _ in range(200):
        if os.path.exists(sock):
            break
        time.sleep(0.02)
    assert os.path.exists(sock)
    yield sock, p
    if p.poll() is None:
        run(["--serve-stop", sock], {})
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    log.close()


def test_requests_run_in_the_server_with_the_clients_world(tmp_path, server):
    sock, proc = server
    d = str(tmp_path)
    lib = write_inputs(d)
    ref = run(["buildlib", lib, os.path.join(d, "ref")], {"MHX_BUILDLIB_HOST": "1"})
    assert ref.returncode == 0
    for i in range(3):  # several requests, one server process
        # relative output path: resolved against the CLIENT's working directory
        p = run(["buildlib", lib, "served%d" % i], {"MHX_BUILDLIB_HOST": "1", "MHX_SERVER": sock}, cwd=d)
        assert p.returncode == 0, p.stderr
        assert "buildlib done" in p.stderr                 # the server's log lines arrive on the client's stderr
        assert canon.digest_file(os.path.join(d, "served%d.bin" % i)) == canon.digest_file(os.path.join(d, "ref.bin"))
        assert open(os.path.join(d, "served%d.lib_info" % i)).read() == open(os.path.join(d, "ref.lib_info")).read()
    assert proc.poll() is None


def test_failure_of_a_request_does_not_end_the_server(tmp_path, server):
    sock, proc = server
    d = str(tmp_path)
    p = run(["buildlib", os.path.join(d, "missing.lib"), os.path.join(d, "x")], {"MHX_BUILDLIB_HOST": "1", "MHX_SERVER": sock})
    assert p.returncode == 1 and "FATAL" in p.stderr
    q = run(["buildlib"], {"MHX_BUILDLIB_HOST": "1", "MHX_SERVER": sock})  # usage error: exit(1) in a process, a failed request here
    assert q.returncode == 1 and "Usage" in q.stderr
    assert proc.poll() is None
    lib = write_inputs(d)
    ok = run(["buildlib", lib, os.path.join(d, "after")], {"MHX_BUILDLIB_HOST": "1", "MHX_SERVER": sock})
    assert ok.returncode == 0


def test_stop_and_fallback_without_a_server(tmp_path, server):
    sock, proc = server
    assert run(["--serve-stop", sock], {}).returncode == 0
    proc.wait(timeout=10)
    assert not os.path.exists(sock)
    d = str(tmp_path)
    lib = write_inputs(d)
    p = run(["buildlib", lib, os.path.join(d, "local")], {"MHX_BUILDLIB_HOST": "1", "MHX_SERVER": sock})  # nobody listens: done by the client
    assert p.returncode == 0 and os.path.exists(os.path.join(d, "local.bin"))


def test_autostart(tmp_path):
    d = str(tmp_path)
    sock = os.path.join(d, "auto.sock")
    lib = write_inputs(d)
    env = {"MHX_BUILDLIB_HOST": "1", "MHX_SERVER": sock, "MHX_SERVER_AUTOSTART": "1", "MHX_SERVE_IDLE_S": "20", "MHX_SERVER_LOG": os.path.join(d, "auto.log")}
    try:
        p = run(["buildlib", lib, os.path.join(d, "a")], env)
        assert p.returncode == 0, p.stderr
        assert os.path.exists(sock)
        q = run(["buildlib", lib, os.path.join(d, "b")], env)
        assert q.returncode == 0
        assert canon.digest_file(os.path.join(d, "a.bin")) == canon.digest_file(os.path.join(d, "b.bin"))
    finally:
        run(["--serve-stop", sock], {})
    for _ in range(100):
        if not os.path.exists(sock):
            break
        time.sleep(0.05)
    assert not os.path.exists(sock)
    assert "leaves after 2 requests" in open(os.path.join(d, "auto.log")).read()


def default_socket(env):
    """where megahit_core's default server listens under this environment (mhx_core --default-socket: one per user, device and
    set of visible devices, in $XDG_RUNTIME_DIR or a 0700 directory of the user's own under /tmp)"""
    return subprocess.run([gu.MHX_CORE, "--default-socket"], env=env, stdout=subprocess.PIPE, text=True, check=True).stdout.strip()


def test_the_references_name_starts_a_server_of_its_own_accord(tmp_path):
    """megahit_core -> mhx_core: no MHX_SERVER in the environment, and the sub-program still runs in a resident server — started by
    this very call, listening on $XDG_RUNTIME_DIR/mhx-core-<uid>-dev0.sock, readable and writable by its owner only; the same
    call with MHX_SERVER=off, or under the name mhx_core, starts nothing"""
    import stat
    d = str(tmp_path)
    link = os.path.join(d, "megahit_core")
    os.symlink(gu.MHX_CORE, link)
    lib = write_inputs(d)
    rt = os.path.join(d, "rt")
    os.mkdir(rt)
    env = {k: v for k, v in os.environ.items() if not k.startswith("MHX_SERVER")}
    env.update(XDG_RUNTIME_DIR=rt, MHX_BUILDLIB_HOST="1", MHX_SERVE_IDLE_S="20")
    sock = default_socket(env)
    assert os.path.dirname(sock) == rt and os.path.basename(sock).startswith("mhx-core-%d-dev0" % os.geteuid())

    def call(prog, out, **extra):
        return subprocess.run([prog, "buildlib", lib, os.path.join(d, out)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(env, **extra), timeout=60)
    try:
        p = call(gu.MHX_CORE, "plain")
        assert p.returncode == 0 and not os.path.exists(sock), p.stderr[-400:]
        p = call(link, "off", MHX_SERVER="off")
        assert p.returncode == 0 and not os.path.exists(sock), p.stderr[-400:]
        p = call(link, "served")
        assert p.returncode == 0, p.stderr[-400:]
        assert os.path.exists(sock), "no server was started"
        assert stat.S_IMODE(os.stat(sock).st_mode) == 0o600
        q = call(link, "served2")
        assert q.returncode == 0
        for name in ("off", "served", "served2"):
            assert canon.digest_file(os.path.join(d, name + ".bin")) == canon.digest_file(os.path.join(d, "plain.bin")), name
    finally:
        if os.path.exists(sock):
            subprocess.run([gu.MHX_CORE, "--serve-stop", sock], timeout=30)


def test_default_socket_without_a_runtime_directory_sits_in_a_directory_of_the_users_own():
    """no $XDG_RUNTIME_DIR (batch jobs): /tmp/mhx-<uid>/ (0700, owned by the caller) instead of a name in /tmp itself, where any local
    user could bind first; jobs with different visible devices get different servers"""
    import stat
    env = {k: v for k, v in os.environ.items() if k != "XDG_RUNTIME_DIR" and not k.endswith("VISIBLE_DEVICES")}
    sock = default_socket(env)
    own = "/tmp/mhx-%d" % os.geteuid()
    assert os.path.dirname(sock) == own and os.path.basename(sock) == "mhx-core-%d-dev0.sock" % os.geteuid()
    st = os.lstat(own)
    assert stat.S_ISDIR(st.st_mode) and st.st_uid == os.geteuid() and stat.S_IMODE(st.st_mode) & 0o077 == 0
    a = default_socket(dict(env, HIP_VISIBLE_DEVICES="3"))
    b = default_socket(dict(env, HIP_VISIBLE_DEVICES="4"))
    assert len({sock, a, b}) == 3 and default_socket(dict(env, HIP_VISIBLE_DEVICES="3")) == a
