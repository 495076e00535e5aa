"""CPU: the per-device admission of persistent jobs (csrc/wn_gate.h, plain C++ -- no HIP in it) compiled with g++ and driven
through a small harness: bookings that fit run side by side, bookings that do not wait for each other -- between threads of one
process and between processes --, jobs of one (process, stream) share a booking, a booking whose owner died is dropped, the wait
is bounded.  The GPU side (two cfg3 jobs on two threads / in two processes finish with oracle-equal output) is tests/test_gpu_parity.py."""
import os
import subprocess
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pytorch-wavenet_amd", "csrc")

HARNESS = r"""
#include "wn_gate.h"
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
// gate <busid> <cap> <need> <hold_ms> <timeout_ms> [n_threads] [same_stream]
//   every thread books `need`, holds it hold_ms, releases; prints per thread "rc waited_ms shared t_in t_out" (ms since start)
//   n_threads < 0: -n_threads threads that start 30 ms apart, thread i books need_i = (i == 1 ? cap - 2 : need): a LARGE job behind a small one
int main(int argc, char** argv) {
    const char* bus = argv[1];
    const int cap = atoi(argv[2]), need = atoi(argv[3]), hold = atoi(argv[4]), tmo = atoi(argv[5]);
    int nt = argc > 6 ? atoi(argv[6]) : 1;
    const int same = argc > 7 ? atoi(argv[7]) : 0;
    const bool staggered = nt < 0;
    if (staggered) nt = -nt;
    if (hold == -2) {  // forge an entry of a LIVE pid (our parent's) with another start time: a re-used pid must not keep a dead owner's booking alive
        WnGateFile tab;
        const int fd = wn_gate_open_locked(wn_gate_path(bus), &tab);
        tab.slot[0].pid = (int32_t)getppid(); tab.slot[0].need = need; tab.slot[0].token = 77; tab.slot[0].born = wn_gate_born((int)getppid()) + 12345;
        tab.slot[0].state = WN_GATE_RUNNING; tab.slot[0].seq = tab.next_seq++;
        wn_gate_close(fd, &tab, true);
        printf("0 0 1\n");
        return 0;
    }
    if (hold == -3) {  // book, release while somebody else holds the FILE lock (the erasure gives up after 2 s), then book the whole device
        std::shared_ptr<WnGateTicket> t; long long w; int sh;
        const int rc1 = wn_gate_acquire(bus, cap, need, nullptr, tmo, &t, &w, &sh);
        printf("booked\n"); fflush(stdout);
        usleep(400 * 1000);
        const long long r0 = wn_gate_now_ms();
        wn_gate_release(t);
        const long long release_ms = wn_gate_now_ms() - r0;
        usleep(1500 * 1000);
        std::shared_ptr<WnGateTicket> t2;
        const int rc2 = wn_gate_acquire(bus, cap, cap, nullptr, tmo, &t2, &w, &sh);
        printf("%d %d %lld %lld\n", rc1, rc2, release_ms, w);
        if (rc2 == 0) wn_gate_release(t2);
        return 0;
    }
    if (hold < 0) {  // "crash": book and exit without releasing
        std::shared_ptr<WnGateTicket> t; long long w; int sh;
        const int rc = wn_gate_acquire(bus, cap, need, nullptr, tmo, &t, &w, &sh);
        printf("%d %lld %d\n", rc, w, sh);
        fflush(stdout);
        _exit(0);
    }
    const long long t0 = wn_gate_now_ms();
    std::vector<std::thread> th;
    std::vector<std::string> lines(nt);
    for (int i = 0; i < nt; ++i)
        th.emplace_back([&, i] {
            std::shared_ptr<WnGateTicket> t; long long w = 0; int sh = -1;
            const void* stream = same ? (const void*)0x10 : (const void*)(uintptr_t)(0x100 + i);
            if (staggered) usleep(30000 * i);
            const int rc = wn_gate_acquire(bus, cap, (staggered && i == 1) ? cap - 2 : need, stream, tmo, &t, &w, &sh);
            const long long tin = wn_gate_now_ms() - t0;
            if (rc == 0) { usleep(hold * 1000); wn_gate_release(t); wn_gate_release(t); }  // (a second release is a no-op)
            char buf[128];
            snprintf(buf, sizeof(buf), "%d %lld %d %lld %lld", rc, w, sh, tin, wn_gate_now_ms() - t0);
            lines[i] = buf;
        });
    for (auto& t : th) t.join();
    for (auto& l : lines) printf("%s\n", l.c_str());
    return 0;
}
"""


@pytest.fixture(scope="module")
def gate(tmp_path_factory):
    d = tmp_path_factory.mktemp("gate")
    src = d / "gate.cpp"
    src.write_text(HARNESS)
    exe = d / "gate"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-Wall", "-I", CSRC, "-o", str(exe), str(src)])
    tabdir = d / "tables"
    tabdir.mkdir()

    def run(*args, wait=True, env_extra=None):
        env = dict(os.environ, WN_GATE_DIR=str(tabdir))
        env.update(env_extra or {})
        p = subprocess.Popen([str(exe)] + [str(a) for a in args], stdout=subprocess.PIPE, text=True, env=env)
        if not wait:
            return p
        out, _ = p.communicate(timeout=60)
        assert p.returncode == 0
        return [[int(v) for v in ln.split()] for ln in out.splitlines()]

    run.collect = lambda p: [[int(v) for v in ln.split()] for ln in p.communicate(timeout=60)[0].splitlines()]
    return run


def test_bookings_that_fit_run_side_by_side(gate):
    rows = gate("0000:aa:00.0", 32, 3, 150, 5000, 4)        # four cfg1-sized jobs: 12 of 32 CUs per XCD
    assert all(r[0] == 0 and r[2] == 1 for r in rows)         # admitted, through the shared table
    assert max(r[3] for r in rows) < 100                      # nobody waited for anybody


def test_bookings_that_do_not_fit_take_turns_between_threads(gate):
    rows = gate("0000:ab:00.0", 32, 28, 200, 5000, 3)        # three cfg3-sized jobs
    assert all(r[0] == 0 for r in rows)
    spans = sorted((r[3], r[4]) for r in rows)
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert b0 >= a1 - 2, spans                            # strictly one after the other
    assert sorted(r[1] for r in rows)[-1] >= 380              # the last one waited for two holds


def test_jobs_of_one_stream_share_a_booking(gate):
    rows = gate("0000:ac:00.0", 32, 28, 200, 5000, 3, 1)     # the same three jobs on ONE stream: the stream serialises them
    assert all(r[0] == 0 for r in rows)
    assert max(r[3] for r in rows) < 100                      # wn_generate stays asynchronous: nobody is held back on the host


def test_two_processes_take_turns_and_a_dead_owner_is_dropped(gate):
    a = gate("0000:ad:00.0", 32, 28, 400, 5000, wait=False)
    time.sleep(0.15)
    b = gate("0000:ad:00.0", 32, 28, 10, 5000)
    ra = gate.collect(a)
    assert ra[0][0] == 0 and b[0][0] == 0
    assert b[0][1] >= 150                                     # the second process waited for the first one's job
    # a process that books and dies: its entry must not close the device
    crash = gate("0000:ae:00.0", 32, 28, -1, 1000)
    assert crash[0][0] == 0
    after = gate("0000:ae:00.0", 32, 28, 10, 2000)
    assert after[0][0] == 0 and after[0][1] < 500
    # ... nor may a stranger that inherited the pid: an entry whose owner's start time is not the live process's is stale
    gate("0000:b1:00.0", 32, 28, -2, 1000)
    again = gate("0000:b1:00.0", 32, 28, 10, 2000)
    assert again[0][0] == 0 and again[0][1] < 500


def test_the_wait_is_bounded(gate):
    a = gate("0000:af:00.0", 32, 28, 800, 5000, wait=False)
    time.sleep(0.15)
    b = gate("0000:af:00.0", 32, 28, 10, 200)                # gives up after 200 ms
    assert b[0][0] == 1 and 200 <= b[0][1] < 700
    assert gate.collect(a)[0][0] == 0


def test_without_a_shared_directory_the_gate_is_process_local(gate):
    rows = gate("0000:b0:00.0", 32, 28, 100, 5000, 2, env_extra={"WN_GATE_DIR": "/nonexistent/dir"})
    assert all(r[0] == 0 and r[2] == 0 for r in rows)         # shared == 0: this process only
    spans = sorted((r[3], r[4]) for r in rows)
    assert spans[1][0] >= spans[0][1] - 2


def test_first_come_first_served_a_large_job_is_not_starved_by_small_ones(gate):
    # thread 0 (small, 10 of 32) runs; thread 1 (LARGE, 30 of 32) arrives 30 ms later and has to wait for it; threads 2..5 (small) arrive behind the
    # large one: each of them would fit next to thread 0, but the large job was there first -- they are admitted behind it, not in front
    rows = gate("0000:b2:00.0", 32, 10, 150, 5000, -6)
    assert all(r[0] == 0 for r in rows)
    large_in, large_out = rows[1][3], rows[1][4]
    assert large_in >= rows[0][4] - 2                          # the large job ran after the first small one ...
    for r in rows[2:]:
        assert r[3] >= large_in - 2, rows                      # ... and nobody that arrived behind it was admitted in front of it
    assert large_in < 400                                       # (it did not wait for all the small ones to come and go)


def test_the_table_is_private_to_the_user_and_never_opened_through_a_link(gate, tmp_path):
    # default location: /dev/shm/wn_mi355_gate_u<euid>_<busid>, mode 0600
    env = {k: v for k, v in os.environ.items() if k != "WN_GATE_DIR"}
    bus = "0000:fe:%02x.0" % (os.getpid() % 200)
    path = "/dev/shm/wn_mi355_gate_u%d_%s" % (os.geteuid(), bus.replace(":", "_").replace(".", "_"))
    try:
        rows = gate(bus, 32, 3, 10, 1000, env_extra={"WN_GATE_DIR": ""})
        assert rows[0][0] == 0 and rows[0][2] == 1
        st = os.stat(path)
        assert (st.st_mode & 0o777) == 0o600 and st.st_uid == os.geteuid()
    finally:
        if os.path.exists(path):
            os.remove(path)
    # a symbolic link planted under the table's name is refused (nothing behind it is created, chmod'ed or written): process-local gate, said out loud
    d = tmp_path / "linkdir"
    d.mkdir()
    victim = tmp_path / "victim"
    victim.write_text("precious")
    os.chmod(victim, 0o644)
    os.symlink(victim, d / "wn_mi355_gate_0000_fd_00_0")
    rows = gate("0000:fd:00.0", 32, 3, 10, 1000, env_extra={"WN_GATE_DIR": str(d)})
    assert rows[0][0] == 0 and rows[0][2] == 0                 # admitted, but NOT through the shared table
    assert victim.read_text() == "precious" and (os.stat(victim).st_mode & 0o777) == 0o644
    # ... and so is a table that everybody may write
    d2 = tmp_path / "worlddir"
    d2.mkdir()
    f = d2 / "wn_mi355_gate_0000_fc_00_0"
    f.write_bytes(b"")
    os.chmod(f, 0o666)
    rows = gate("0000:fc:00.0", 32, 3, 10, 1000, env_extra={"WN_GATE_DIR": str(d2)})
    assert rows[0][0] == 0 and rows[0][2] == 0
    assert f.read_bytes() == b""


def test_a_release_that_cannot_get_the_file_lock_is_retried_by_the_next_booking(gate, tmp_path):
    # Somebody holds the table's lock (a stopped process, say) while a job's booking is released: the erasure gives up after 2 s instead of hanging
    # the HIP callback thread -- and the entry of this LIVE process (which nobody else may drop) is swept by the process's next visit to the table.
    import fcntl
    d = tmp_path / "orphans"
    d.mkdir()
    env = {"WN_GATE_DIR": str(d)}
    p = gate("0000:fb:00.0", 32, 20, -3, 6000, wait=False, env_extra=env)
    assert p.stdout.readline().strip() == "booked"
    with open(d / "wn_mi355_gate_0000_fb_00_0", "r+b") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        time.sleep(2.8)                                        # the release (0.4 s after the booking) tries for 2 s and gives up
        fcntl.flock(f, fcntl.LOCK_UN)
    rc1, rc2, release_ms, waited = [int(v) for v in p.communicate(timeout=30)[0].split()]
    assert rc1 == 0 and 1900 <= release_ms <= 2600
    assert rc2 == 0 and waited < 1000                          # the whole device could be booked: the orphaned 20 were swept first
