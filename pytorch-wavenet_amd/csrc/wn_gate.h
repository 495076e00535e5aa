// wn_gate.h -- per-device admission of persistent generation jobs (host only).
//
// A generation job is ONE persistent kernel whose workgroups each own a CU for the whole job and wait for each other through
// hand-off granules: it only makes progress once ALL of them are resident.  Two such jobs that do not fit the chip together (cfg3:
// 220 workgroups of 256 CUs each) launched from two threads or two processes can each get part of the chip -- neither becomes
// resident, both spin until their hand-off timeouts (WN_E_TIMEOUT).  The reference calls generate_fast from a daemon thread next to
// the training loop (/root/reference/model_logging.py:48-58), so this is a use the drop-in has to survive.
//
// The gate: every job books the CUs it needs PER XCD (blocks are dispatched round-robin over the 8 XCDs: a full XCD stalls the
// dispatch however empty the others are) in a small table shared by the processes that use the device -- a file named after the
// device's PCI bus id, read and written under flock().  Admission is FIRST COME FIRST SERVED: a job that does not fit at once takes
// a WAITING entry with a sequence number; a job is admitted when its booking fits next to the running ones AND no job with a smaller
// sequence number is still waiting (a stream of small jobs cannot starve a large one).  The wait is bounded (timeout_ms covers
// everything: the file lock, the jobs in front).  Entries carry the owner's pid and start time; entries of processes that no longer
// exist (or whose pid now belongs to another process) are dropped, so a crashed owner cannot close the device for everyone else.
// Jobs that a HIP stream already serialises (same process, same stream: the rounds of a large job, back-to-back calls of one caller)
// share ONE booking -- the launch stays asynchronous there.  The booking is released by a host function enqueued behind the kernel
// (hipLaunchHostFunc), at the latest by wn_wait.
// Locks: the registry mutex (threads of this process) is never held while the FILE lock is awaited; the file lock is only ever tried, with a
// bound (a job's own timeout when it books, 2 s when it releases -- an erasure that gives up leaves an orphan this process sweeps on its next
// visit to the table: an entry of a live process is never dropped by anybody else).
//
// Where the table lives (round 5, after the advisor's review of the round-4 file -- world-writable, created through O_CREAT in /dev/shm):
//   default          /dev/shm/wn_mi355_gate_u<euid>_<busid>, mode 0600: the processes of ONE user coordinate.  Jobs of different users
//                    on one device are not serialised against each other (they never were before the gate existed; a table every local
//                    user may write is a table every local user may fill or redirect).
//   WN_GATE_DIR=dir  <dir>/wn_mi355_gate_<busid>, mode 0660 & ~umask: a directory an administrator prepared for the users that share the
//                    device (e.g. group-writable, setgid) makes them share one table.
// The file is opened O_NOFOLLOW, created only with O_CREAT | O_EXCL | O_NOFOLLOW, and trusted only after fstat(): a regular file with one
// link, owned by this user (default location) and not writable by "other".  Anything else is an ERROR the first booking reports once on
// stderr, and the device's gate is process-local from then on: a device keeps ONE mode (shared or local) for the life of the process, a
// transient failure to open the table later is a retry, never a silent change of mode.
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
#include <vector>

#define WN_GATE_SLOTS 64
// need: CUs per XCD; pid 0 = free; born: the owner's start time; seq: order of arrival; state: WN_GATE_RUNNING / WN_GATE_WAITING
struct WnGateSlot { int32_t pid; int32_t need; int64_t token; int64_t born; int64_t seq; int32_t state; int32_t pad; };
struct WnGateFile { uint32_t magic, version; int64_t next_seq; WnGateSlot slot[WN_GATE_SLOTS]; };
#define WN_GATE_MAGIC 0x474e5733u  /* "3WNG" */
#define WN_GATE_VERSION 3u
#define WN_GATE_RUNNING 1
#define WN_GATE_WAITING 2

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
    std::map<std::string, int> mode;                                    // device -> 1 shared table, 0 process-local: decided ONCE per device
    std::map<std::string, int64_t> local_next, local_serving;           // process-local first come first served
    std::multimap<std::string, int64_t> orphans;                        // table path -> tokens of OUR entries whose erasure did not get the file lock in time
    int64_t next_token = 1;
};
// heap-allocated and never destroyed: the runtime's callback thread (hipLaunchHostFunc -> wn_gate_release) may still run while the
// process's static destructors do
static inline WnGateRegistry& wn_gate_registry() { static WnGateRegistry* r = new WnGateRegistry(); return *r; }

static inline std::string wn_gate_path(const char* busid) {
    std::string name;
    for (const char* c = busid; *c; ++c) name += ((*c >= '0' && *c <= '9') || (*c >= 'a' && *c <= 'z') || (*c >= 'A' && *c <= 'Z')) ? *c : '_';
    const char* dir = getenv("WN_GATE_DIR");
    if (dir && dir[0]) return std::string(dir) + "/wn_mi355_gate_" + name;
    char uid[32];
    snprintf(uid, sizeof(uid), "u%u_", (unsigned)geteuid());
    return std::string("/dev/shm/wn_mi355_gate_") + uid + name;
}

static inline long long wn_gate_now_ms() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 1000 + ts.tv_nsec / 1000000;
}

// Opens the table without following links, creating it exclusively when it does not exist, and checks what was opened.
// Returns the descriptor; -1 with *why set (a static text) otherwise.  Nothing is ever chmod'ed or written before the checks have passed.
static inline int wn_gate_open_checked(const std::string& path, const char** why) {
    const bool admin_dir = getenv("WN_GATE_DIR") && getenv("WN_GATE_DIR")[0];
    const mode_t mode = admin_dir ? 0660 : 0600;
    int fd = -1;
    for (int attempt = 0; attempt < 4 && fd < 0; ++attempt) {
        fd = open(path.c_str(), O_RDWR | O_NOFOLLOW | O_CLOEXEC);
        if (fd >= 0 || errno != ENOENT) break;
        fd = open(path.c_str(), O_RDWR | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, mode);
        if (fd < 0 && errno != EEXIST) break;   // (EEXIST: somebody else created it between the two calls -- open it)
    }
    if (fd < 0) {
        *why = errno == ELOOP ? "the table's name is a symbolic link" : errno == EACCES ? "no permission to open the table (another user's file?)" : "cannot open or create the table";
        return -1;
    }
    struct stat st;
    if (fstat(fd, &st) != 0) { *why = "fstat failed"; close(fd); return -1; }
    if (!S_ISREG(st.st_mode)) { *why = "the table is not a regular file"; close(fd); return -1; }
    if (st.st_nlink != 1) { *why = "the table has more than one link"; close(fd); return -1; }
    if (!admin_dir && st.st_uid != geteuid()) { *why = "the table belongs to another user"; close(fd); return -1; }
    if (st.st_mode & S_IWOTH) { *why = "the table is writable by everybody"; close(fd); return -1; }
    if (st.st_size != 0 && st.st_size != (off_t)sizeof(WnGateFile)) { *why = "the table has an unexpected size"; close(fd); return -1; }
    return fd;
}

// Opens and locks the table (the lock is tried without blocking until deadline_ms: a stopped process that holds it cannot stall
// the caller beyond its own timeout); returns the descriptor, -1 = could not (transient or not: *why says), -2 = deadline.
static inline int wn_gate_open_locked(const std::string& path, WnGateFile* tab, long long deadline_ms = -1, const char** why = nullptr) {
    const char* dummy = nullptr;
    if (!why) why = &dummy;
    const int fd = wn_gate_open_checked(path, why);
    if (fd < 0) return -1;
    long long backoff_us = 50;
    while (flock(fd, LOCK_EX | LOCK_NB) != 0) {
        if (errno != EWOULDBLOCK && errno != EINTR) { *why = "flock failed"; close(fd); return -1; }
        if (deadline_ms >= 0 && wn_gate_now_ms() > deadline_ms) { *why = "the table's lock stayed taken"; close(fd); return -2; }
        usleep((useconds_t)backoff_us);
        if (backoff_us < 2000) backoff_us *= 2;
    }
    const ssize_t n = pread(fd, tab, sizeof(*tab), 0);
    if (n != (ssize_t)sizeof(*tab) || tab->magic != WN_GATE_MAGIC || tab->version != WN_GATE_VERSION) {
        memset(tab, 0, sizeof(*tab));
        tab->magic = WN_GATE_MAGIC; tab->version = WN_GATE_VERSION; tab->next_seq = 1;
    }
    return fd;
}
// drops the entries of processes that are gone (called only when a booking does not fit: kill() + a /proc read per entry)
static inline bool wn_gate_drop_stale(WnGateFile* tab) {
    bool dropped = false;
    for (int i = 0; i < WN_GATE_SLOTS; ++i) {
        WnGateSlot& s = tab->slot[i];
        if (s.pid <= 0) continue;
        const bool gone = kill((pid_t)s.pid, 0) != 0 && errno == ESRCH;
        const int64_t born = gone ? 0 : wn_gate_born(s.pid);   // (0: /proc not readable -- keep the entry, the pid answers)
        if (gone || (born != 0 && s.born != 0 && born != s.born)) { memset(&s, 0, sizeof(s)); dropped = true; }
    }
    return dropped;
}
static inline void wn_gate_close(int fd, const WnGateFile* tab, bool dirty) {
    if (dirty) (void)!pwrite(fd, tab, sizeof(*tab), 0);
    (void)flock(fd, LOCK_UN);
    close(fd);
}

// one admission attempt in the (locked) table for the entry (pid, token): 1 admitted, 0 waiting (a WAITING entry is kept), -1 table full
static inline int wn_gate_try_table(WnGateFile* tab, int cap, int need, int64_t token, int64_t born, bool* dirty) {
    const int32_t me = (int32_t)getpid();
    for (int pass = 0; pass < 2; ++pass) {
        int used = 0, free_slot = -1, mine = -1;
        int64_t oldest_waiting = INT64_MAX;
        for (int i = 0; i < WN_GATE_SLOTS; ++i) {
            const WnGateSlot& s = tab->slot[i];
            if (s.pid <= 0) { if (free_slot < 0) free_slot = i; continue; }
            if (s.pid == me && s.token == token) { mine = i; continue; }
            if (s.state == WN_GATE_RUNNING) used += s.need;
            else if (s.seq < oldest_waiting) oldest_waiting = s.seq;
        }
        const int64_t my_seq = mine >= 0 ? tab->slot[mine].seq : tab->next_seq;
        if (used + need <= cap && my_seq < oldest_waiting && (mine >= 0 || free_slot >= 0)) {
            WnGateSlot& s = tab->slot[mine >= 0 ? mine : free_slot];
            if (mine < 0) { s.pid = me; s.need = need; s.token = token; s.born = born; s.seq = tab->next_seq++; s.pad = 0; }
            s.state = WN_GATE_RUNNING;
            *dirty = true;
            return 1;
        }
        if (pass == 0 && wn_gate_drop_stale(tab)) { *dirty = true; continue; }   // something in front of us may be dead: look again
        if (mine < 0) {
            if (free_slot < 0) return -1;
            WnGateSlot& s = tab->slot[free_slot];
            s.pid = me; s.need = need; s.token = token; s.born = born; s.seq = tab->next_seq++; s.state = WN_GATE_WAITING; s.pad = 0;
            *dirty = true;
        }
        return 0;
    }
    return 0;
}
// Erases this process's entry `token` (and the entries in `also`: earlier erasures that did not get the lock) from the table.  false: the lock
// stayed taken for 2 s or the table could not be opened -- the caller keeps the token as an orphan and the next visit to the table retries
// (an entry of a LIVE process is never dropped by anybody else: left behind, it would book CUs for the life of this process).
static inline bool wn_gate_erase_entry(const std::string& path, int64_t token, const int64_t* also = nullptr, int n_also = 0) {
    WnGateFile tab;
    const int fd = wn_gate_open_locked(path, &tab, wn_gate_now_ms() + 2000);
    if (fd < 0) return false;
    const int32_t me = (int32_t)getpid();
    bool dirty = false;
    for (int i = 0; i < WN_GATE_SLOTS; ++i) {
        WnGateSlot& sl = tab.slot[i];
        if (sl.pid != me) continue;
        bool hit = sl.token == token;
        for (int k = 0; k < n_also && !hit; ++k) hit = sl.token == also[k];
        if (hit) { memset(&sl, 0, sizeof(sl)); dirty = true; }
    }
    wn_gate_close(fd, &tab, dirty);
    return true;
}
// (both below: under no lock of ours while the FILE lock is awaited -- see wn_gate_acquire)
static inline void wn_gate_erase_or_remember(const std::string& path, int64_t token);

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
    const long long t0 = wn_gate_now_ms(), deadline = t0 + timeout_ms;
    *waited_ms = 0;
    int64_t token = 0, local_ticket = 0;   // this job's identity in the table / its place in the process-local queue (0: none yet)
    const int64_t born = wn_gate_born((int)getpid());
    long long nap_us = 200;
    auto give_up = [&]() {
        if (token) {
            bool in_table;
            { std::lock_guard<std::mutex> g(reg.mu); auto m = reg.mode.find(dev); in_table = m == reg.mode.end() || m->second == 1; }   // (undecided: a WAITING entry may exist)
            if (in_table) wn_gate_erase_or_remember(path, token);
        }
        if (local_ticket) {   // leave the process-local queue: whoever is behind us must not wait for a job that will never run
            std::lock_guard<std::mutex> g(reg.mu);
            if (reg.local_serving[dev] == local_ticket) reg.local_serving[dev]++;
            else reg.local_used[dev + "|left|" + std::to_string(local_ticket)] = 1;
        }
        *waited_ms = wn_gate_now_ms() - t0;
        return 1;
    };
    // The registry mutex is never held while the FILE lock is awaited (a stopped process that holds the file lock would otherwise stall every
    // thread of this process that books or releases -- the HIP runtime's callback thread among them -- for this job's whole timeout).
    for (;;) {
        int mode;
        bool joined = false, maybe_in_table = false;
        std::vector<int64_t> sweep;   // our orphaned entries in this table: erased on this visit
        {
            std::lock_guard<std::mutex> g(reg.mu);
            auto it = reg.by_stream.find(key);
            if (it != reg.by_stream.end() && it->second->jobs > 0 && it->second->need >= need) {
                // a job of this process on this very stream is in flight: the stream serialises us behind it, one booking covers both
                // (also when we already queued -- a sibling that started booking at the same moment got there first: leave the queue)
                it->second->jobs++;
                auto t = std::make_shared<WnGateTicket>();
                t->booking = it->second;
                *out = t; *shared = it->second->path.empty() ? 0 : 1;
                if (local_ticket) {
                    if (reg.local_serving[dev] == local_ticket) reg.local_serving[dev]++;
                    else reg.local_used[dev + "|left|" + std::to_string(local_ticket)] = 1;
                }
                auto m = reg.mode.find(dev);
                maybe_in_table = token != 0 && (m == reg.mode.end() || m->second == 1);
                joined = true;
            }
        }
        if (joined) {
            if (maybe_in_table) wn_gate_erase_or_remember(path, token);   // (a WAITING entry we may have left in the table; no entry: a no-op visit)
            *waited_ms = wn_gate_now_ms() - t0;
            return 0;
        }
        {
            std::lock_guard<std::mutex> g(reg.mu);
            // (a larger job behind a smaller one on the same stream books on its own: conservative, never wrong)
            if (!token) token = reg.next_token++;
            auto mode_it = reg.mode.find(dev);
            mode = mode_it == reg.mode.end() ? -1 : mode_it->second;
            if (mode != 0) {
                auto range = reg.orphans.equal_range(path);
                for (auto o = range.first; o != range.second; ++o) sweep.push_back(o->second);
            }
        }
        bool ok = false;
        if (mode != 0) {
            WnGateFile tab;
            const char* why = "";
            const int fd = wn_gate_open_locked(path, &tab, deadline, &why);
            if (fd >= 0) {
                bool dirty = false;
                const int32_t me = (int32_t)getpid();
                for (int i = 0; i < WN_GATE_SLOTS && !sweep.empty(); ++i)
                    for (int64_t tk : sweep)
                        if (tab.slot[i].pid == me && tab.slot[i].token == tk) { memset(&tab.slot[i], 0, sizeof(tab.slot[i])); dirty = true; }
                const int r = wn_gate_try_table(&tab, cap, need, token, born, &dirty);
                wn_gate_close(fd, &tab, dirty);
                ok = r == 1;
                std::lock_guard<std::mutex> g(reg.mu);
                if (mode < 0) { auto ins = reg.mode.emplace(dev, 1); mode = ins.first->second; }   // (another thread may have decided meanwhile: its word stands)
                for (int64_t tk : sweep) {
                    auto range = reg.orphans.equal_range(path);
                    for (auto o = range.first; o != range.second; ++o) if (o->second == tk) { reg.orphans.erase(o); break; }
                }
                if (mode == 0 && ok) {   // (lost the race against a thread that found the table unusable: hand the entry back, queue locally)
                    ok = false;
                }
            } else if (mode < 0 && fd == -1) {
                // the FIRST booking on this device decides its mode for the life of the process -- and says why out loud
                std::lock_guard<std::mutex> g(reg.mu);
                auto ins = reg.mode.emplace(dev, 0);
                mode = ins.first->second;
                if (ins.second)
                    fprintf(stderr, "wn_mi355: admission table %s unusable (%s): the jobs of THIS process are serialised on device %s, other "
                            "processes are not (set WN_GATE_DIR to a directory the device's users share)\n", path.c_str(), why, busid);
            }   // (mode == 1 and the table cannot be opened right now, or the lock is held: not admitted yet -- retry below)
            if (mode == 0 && fd >= 0) wn_gate_erase_or_remember(path, token);
        }
        {
            std::lock_guard<std::mutex> g(reg.mu);
            if (mode == 0) {   // process-local, first come first served
                if (!local_ticket) { int64_t& n = reg.local_next[dev]; if (n == 0) { n = 1; reg.local_serving[dev] = 1; } local_ticket = n++; }
                int64_t& serving = reg.local_serving[dev];
                while (serving < local_ticket && reg.local_used.erase(dev + "|left|" + std::to_string(serving))) ++serving;   // (tickets that gave up)
                int& used = reg.local_used[dev];
                if (serving == local_ticket && used + need <= cap) { used += need; ++serving; ok = true; }
            }
            if (ok) {
                auto b = std::make_shared<WnGateBooking>();
                b->key = key; b->need = need; b->jobs = 1; b->token = token;
                if (mode == 1) b->path = path;
                *shared = mode;
                auto it = reg.by_stream.find(key);
                if (it == reg.by_stream.end() || it->second->jobs == 0) reg.by_stream[key] = b;  // (joinable by later jobs of this stream)
                auto t = std::make_shared<WnGateTicket>();
                t->booking = b;
                *out = t;
                *waited_ms = wn_gate_now_ms() - t0;
                return 0;
            }
            *shared = mode < 0 ? 0 : mode;
        }
        if (wn_gate_now_ms() > deadline) return give_up();
        usleep((useconds_t)nap_us);
        if (nap_us < 2000) nap_us *= 2;   // 0.2 ms ... 2 ms between looks: a cfg3 job is tens of milliseconds at the least
    }
}

static inline void wn_gate_erase_or_remember(const std::string& path, int64_t token) {
    WnGateRegistry& reg = wn_gate_registry();
    std::vector<int64_t> also;
    {
        std::lock_guard<std::mutex> g(reg.mu);
        auto range = reg.orphans.equal_range(path);
        for (auto o = range.first; o != range.second; ++o) also.push_back(o->second);
    }
    const bool done = wn_gate_erase_entry(path, token, also.data(), (int)also.size());
    std::lock_guard<std::mutex> g(reg.mu);
    if (done) {
        for (int64_t tk : also) {
            auto range = reg.orphans.equal_range(path);
            for (auto o = range.first; o != range.second; ++o) if (o->second == tk) { reg.orphans.erase(o); break; }
        }
    } else {
        reg.orphans.emplace(path, token);
    }
}

static inline void wn_gate_release(const std::shared_ptr<WnGateTicket>& t) {
    if (!t || t->released.exchange(1) != 0) return;
    WnGateRegistry& reg = wn_gate_registry();
    std::string path;
    int64_t token = 0;
    {
        std::lock_guard<std::mutex> g(reg.mu);
        WnGateBooking& b = *t->booking;
        if (--b.jobs > 0) return;
        auto it = reg.by_stream.find(b.key);
        if (it != reg.by_stream.end() && it->second.get() == &b) reg.by_stream.erase(it);
        if (!b.path.empty()) { path = b.path; token = b.token; }
        else {
            const std::string dev = b.key.substr(0, b.key.find('|'));
            int& used = reg.local_used[dev];
            used -= b.need;
            if (used < 0) used = 0;
        }
    }
    // (outside the registry mutex: this may run on the HIP runtime's callback thread, and the file lock is only ever tried, with a bound)
    if (!path.empty()) wn_gate_erase_or_remember(path, token);
}

#endif  // WN_GATE_H
