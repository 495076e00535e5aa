// wn_gate.h -- per-device admission of persistent generation jobs (host only).
//
// A generation job is ONE persistent kernel whose workgroups each own a CU for the whole job and wait for each other through
// hand-off granules: it only makes progress once ALL of them are resident.  Two such jobs that do not fit the chip together (cfg3:
// 220 workgroups of 256 CUs each) launched from two threads or two processes can each get part of the chip -- neither becomes
// resident, both spin until their hand-off timeouts (WN_E_TIMEOUT).  The reference calls generate_fast from a daemon thread next to
// the training loop (/root/reference/model_logging.py:48-58), so this is a use the drop-in has to survive.
//
// The gate: every job books the CUs it needs PER XCD (blocks are dispatched round-robin over the 8 XCDs, so a job of n blocks
// needs ceil(n / 8) CUs on every XCD, and a full XCD stalls the dispatch however empty the others are) in a small table shared by
// all processes that use the device -- a file in /dev/shm named after the device's PCI bus id, read and written under flock().
// A job is admitted when its booking fits next to the bookings already there; otherwise wn_generate WAITS (bounded) until the
// jobs in front of it have finished.  Bookings carry the owner's pid and start time; entries of processes that no longer exist (or whose
// pid now belongs to another process) are dropped, so a crashed owner cannot close the device for everyone else.  Jobs that a HIP stream already serialises (same process, same
// stream: the rounds of a large job, back-to-back calls of one caller) share ONE booking -- the launch stays asynchronous there.
// The booking is released by a host function enqueued behind the kernel (hipLaunchHostFunc), at the latest by wn_wait.
//
// No /dev/shm (or no permission): the gate still serialises the threads of this process, and says so in wn_get_info.
#ifndef WN_GATE_H
#define WN_GATE_H

#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>

#define WN_GATE_SLOTS 64
struct WnGateSlot { int32_t pid; int32_t need; int64_t token; int64_t born; };   // need: CUs per XCD; pid 0 = free; born: the owner's start time
struct WnGateFile { uint32_t magic, version; WnGateSlot slot[WN_GATE_SLOTS]; };
#define WN_GATE_MAGIC 0x474e5732u  /* "2WNG" */

// Start time of a process in clock ticks since boot (/proc/<pid>/stat, field 22), 0 if it does not exist: a pid alone does not identify the
// owner of a booking -- pids are re-used, and an entry left behind by a process that died mid-job must not be kept alive by a stranger.
static inline int64_t wn_gate_born(int pid) {
    char path[64], buf[1024];
    snprintf(path, sizeof(path), "/proc/%d/stat", pid);
    FILE* f = fopen(path, "r");
    if (!f) return 0;
    const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
    fclose(f);
    buf[n] = 0;
    const char* p = strrchr(buf, ')');   // (the command name may contain spaces and parentheses: fields resume behind the LAST one)
    if (!p) return 0;
    long long v = 0;
    int field = 2;
    for (p += 1; *p && field < 22; ++p)
        if (*p == ' ' && p[1] != ' ') ++field;
    if (field == 22 && sscanf(p, "%lld", &v) == 1 && v > 0) return (int64_t)v;
    return 0;
}

struct WnGateBooking {           // one booking in the shared table; shared by the jobs of one (process, stream)
    std::string path;            // "" = process-local only
    std::string key;             // device + stream (registry key)
    int64_t token = 0;
    int need = 0;
    int jobs = 0;                // jobs in flight that ride on this booking (guarded by the registry mutex)
};
struct WnGateTicket {            // one job's share of a booking; released exactly once (host function or wn_wait, whichever is first)
    std::shared_ptr<WnGateBooking> booking;
    std::atomic<int> released{0};
};

struct WnGateRegistry {
    std::mutex mu;
    std::map<std::string, std::shared_ptr<WnGateBooking>> by_stream;   // live bookings of this process
    std::map<std::string, int> local_used;                              // device -> CUs per XCD booked by this process (no shared file)
    int64_t next_token = 1;
};
static inline WnGateRegistry& wn_gate_registry() { static WnGateRegistry r; return r; }

static inline std::string wn_gate_path(const char* busid) {
    std::string name = "wn_mi355_gate_";
    for (const char* c = busid; *c; ++c) name += ((*c >= '0' && *c <= '9') || (*c >= 'a' && *c <= 'z') || (*c >= 'A' && *c <= 'Z')) ? *c : '_';
    const char* dir = getenv("WN_GATE_DIR");
    return std::string(dir && dir[0] ? dir : "/dev/shm") + "/" + name;
}

// Opens (creating if needed) and locks the table; returns the descriptor or -1.
static inline int wn_gate_open_locked(const std::string& path, WnGateFile* tab) {
    const int fd = open(path.c_str(), O_RDWR | O_CREAT | O_CLOEXEC, 0666);
    if (fd < 0) return -1;
    (void)fchmod(fd, 0666);  // (other users of the same device must be able to book as well; the umask may have cut the mode)
    if (flock(fd, LOCK_EX) != 0) { close(fd); return -1; }
    const ssize_t n = pread(fd, tab, sizeof(*tab), 0);
    if (n != (ssize_t)sizeof(*tab) || tab->magic != WN_GATE_MAGIC || tab->version != 2) {
        memset(tab, 0, sizeof(*tab));
        tab->magic = WN_GATE_MAGIC; tab->version = 2;
    }
    for (int i = 0; i < WN_GATE_SLOTS; ++i) {  // bookings of processes that are gone
        WnGateSlot& s = tab->slot[i];
        if (s.pid <= 0) continue;
        const bool gone = kill((pid_t)s.pid, 0) != 0 && errno == ESRCH;
        const int64_t born = gone ? 0 : wn_gate_born(s.pid);   // (0: /proc not readable -- keep the entry, the pid answers)
        if (gone || (born != 0 && s.born != 0 && born != s.born)) memset(&s, 0, sizeof(s));
    }
    return fd;
}
static inline void wn_gate_close(int fd, const WnGateFile* tab, bool dirty) {
    if (dirty) (void)!pwrite(fd, tab, sizeof(*tab), 0);
    (void)flock(fd, LOCK_UN);
    close(fd);
}

static inline long long wn_gate_now_ms() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 1000 + ts.tv_nsec / 1000000;
}

// Books `need` CUs per XCD (of `cap`) on the device `busid` for a job on `stream`.  Returns 0 and a ticket; 1 when the wait ran
// into timeout_ms (nothing booked).  *waited_ms: how long the job was held back; *shared: 1 = inter-process table, 0 = this process only.
static inline int wn_gate_acquire(const char* busid, int cap, int need, const void* stream, long long timeout_ms, std::shared_ptr<WnGateTicket>* out,
                                  long long* waited_ms, int* shared) {
    WnGateRegistry& reg = wn_gate_registry();
    if (need > cap) need = cap;  // (a job larger than the device fails at launch, not here)
    char skey[64];
    snprintf(skey, sizeof(skey), "|%p", stream);
    const std::string dev(busid), key = dev + skey;
    const std::string path = wn_gate_path(busid);
    const long long t0 = wn_gate_now_ms();
    *waited_ms = 0;
    for (;;) {
        {
            std::lock_guard<std::mutex> g(reg.mu);
            auto it = reg.by_stream.find(key);
            if (it != reg.by_stream.end() && it->second->jobs > 0 && it->second->need >= need) {
                // a job of this process on this very stream is in flight: the stream serialises us behind it, one booking covers both
                it->second->jobs++;
                auto t = std::make_shared<WnGateTicket>();
                t->booking = it->second;
                *out = t; *shared = it->second->path.empty() ? 0 : 1;
                return 0;
            }
            // (a larger job behind a smaller one on the same stream books on its own: conservative, never wrong)
            WnGateFile tab;
            const int fd = wn_gate_open_locked(path, &tab);
            bool ok = false;
            auto b = std::make_shared<WnGateBooking>();
            b->key = key; b->need = need; b->jobs = 1; b->token = reg.next_token++;
            if (fd >= 0) {
                int used = 0, free_slot = -1;
                for (int i = 0; i < WN_GATE_SLOTS; ++i) {
                    if (tab.slot[i].pid > 0) used += tab.slot[i].need;
                    else if (free_slot < 0) free_slot = i;
                }
                if (used + need <= cap && free_slot >= 0) {
                    tab.slot[free_slot].pid = (int32_t)getpid(); tab.slot[free_slot].need = need; tab.slot[free_slot].token = b->token;
                    tab.slot[free_slot].born = wn_gate_born((int)getpid());
                    b->path = path;
                    ok = true;
                }
                wn_gate_close(fd, &tab, ok);
                *shared = 1;
            } else {
                int& used = reg.local_used[dev];
                if (used + need <= cap) { used += need; ok = true; }
                *shared = 0;
            }
            if (ok) {
                if (it == reg.by_stream.end() || it->second->jobs == 0) reg.by_stream[key] = b;  // (joinable by later jobs of this stream)
                auto t = std::make_shared<WnGateTicket>();
                t->booking = b;
                *out = t;
                *waited_ms = wn_gate_now_ms() - t0;
                return 0;
            }
        }
        if (wn_gate_now_ms() - t0 > timeout_ms) { *waited_ms = wn_gate_now_ms() - t0; return 1; }
        usleep(200);
    }
}

static inline void wn_gate_release(const std::shared_ptr<WnGateTicket>& t) {
    if (!t || t->released.exchange(1) != 0) return;
    WnGateRegistry& reg = wn_gate_registry();
    std::lock_guard<std::mutex> g(reg.mu);
    WnGateBooking& b = *t->booking;
    if (--b.jobs > 0) return;
    auto it = reg.by_stream.find(b.key);
    if (it != reg.by_stream.end() && it->second.get() == &b) reg.by_stream.erase(it);
    if (!b.path.empty()) {
        WnGateFile tab;
        const int fd = wn_gate_open_locked(b.path, &tab);
        if (fd >= 0) {
            const int32_t me = (int32_t)getpid();
            for (int i = 0; i < WN_GATE_SLOTS; ++i)
                if (tab.slot[i].pid == me && tab.slot[i].token == b.token) memset(&tab.slot[i], 0, sizeof(tab.slot[i]));
            wn_gate_close(fd, &tab, true);
        }
    } else {
        const std::string dev = b.key.substr(0, b.key.find('|'));
        int& used = reg.local_used[dev];
        used -= b.need;
        if (used < 0) used = 0;
    }
}

#endif  // WN_GATE_H
