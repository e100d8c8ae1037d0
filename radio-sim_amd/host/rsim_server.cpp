// rsim_server.cpp -- the radio-link server: the reference's JSON/TCP front end
// re-stated natively, with the MI355X medium behind it.  One thread, one poll() loop; a whole simulation tick is
// ONE evaluation on the device (tick mode) and the reception state machine, the ordered deliveries and the
// node-info of the time-step messages come from the device's event stage (radiomedium.hpp, rm_events_*).
//
// What it restates (paths under /root/reference/radio-medium/java/se/sics/emul8/radiomedium/):
//   net/Server.java:57-126                 port 7711, the greeting {"radio-simulator":{"name":"RSIM 0.1",
//                                          "api-version":"0.6"},"status":"OK"} sent to every new connection
//   net/JSONClientConnection.java:134-255  framing: a '{' starts brace counting (quotes and backslashes honoured,
//                                          CR dropped outside strings); otherwise lines "<size>[;attr=..]": size > 0
//                                          reads that many bytes of UTF-8 JSON, 0 nothing, < 0 brace counting
//   net/JSONClientConnection.java:257-287  send: the minimal JSON text + CR LF
//   net/JSONClientConnection.java:326-434  time-step / time-step-done / receive / event messages
//   net/SimulatorJSONHandler.java:28-273   commands time-get, time-set, transmit, log, node-config-set,
//                                          link-quality, configuration-set, subscribe-event, unsubscribe-event;
//                                          reply / error-reply objects and when they are sent
//   Simulator.java:118-194,249-277,312-364 message ids (1001, 1002, ...), stepTime / emulatorTimeStepped /
//                                          emulatorTimeStepDone, addNode and the emulator list, event listeners,
//                                          deliverRadioPacket
//   Main.java:46-86                        -pcap; the null radio medium is the default
//
// Behaviour kept on purpose, because an emulator written against the reference sees it:
//   * an exception inside the reader (bad JSON, a member of the wrong type, "log" for an unknown node) ends that
//     connection; a closed connection stays the time controller / an emulator (the reference never removes it),
//     so a step that waits for it never finishes;
//   * node ids are the JSON text of the "node-id" value (1 -> "1", "n1" -> "\"n1\"");
//   * a byte >= 0x80 read in brace-counting mode is one Latin-1 character (written back as two UTF-8 bytes).
//
// Differences, all on the host side of the medium:
//   * one thread: messages are handled in the order poll() returns them, not by one thread per connection;
//   * "transmit" in tick mode is queued and evaluated at the end of the step (or before the next command that
//     changes a node or the medium), which gives the same calls in the same order (tests/test_gpu_host_tick.py);
//   * no web server (-ws), no logback: --verbose prints one line per message to stderr.
//
// There is no CPU evaluation of the medium here: without a gfx950 device the server refuses to start unless
// --no-medium is given, and then "transmit" answers "no radio medium available" exactly as the reference does
// when Simulator.getRadioMedium() is null (protocol tests only).
#include <arpa/inet.h>
#include <cerrno>
#include <chrono>
#include <csignal>
#include <cstring>
#include <fcntl.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <sys/uio.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <cstdio>
#include <deque>
#include <unordered_map>
#include <memory>
#include <string>
#include <vector>

#include "json.hpp"
#include "radiomedium.hpp"

namespace rsim {

using emul8::GpuRadioMedium;
using emul8::Node;
using emul8::RadioPacket;

static bool g_verbose = false;
#define VLOG(...) do { if (g_verbose) { std::fprintf(stderr, __VA_ARGS__); std::fputc('\n', stderr); } } while (0)

// The reference serialises on a thread per connection (net/JSONClientConnection.java:118-131: every connection's reader thread
// writes its own replies).  Here one thread owns the protocol state, and the two jobs whose volume grows with the simulation --
// rewriting the changed nodes' node-info objects, writing a step's receive messages -- are cut into independent parts and
// handed to a few workers: run(parts, fn) calls fn(part) for every part on the pool (the caller takes parts too) and returns
// when all are done.  Nothing else runs while a job does: the workers touch only what their part owns.
class Workers {
public:
    explicit Workers(unsigned n)
    {
        for (unsigned i = 0; i + 1 < n; ++i) threads_.emplace_back([this] { loop(); });
    }
    ~Workers()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    unsigned size() const { return unsigned(threads_.size()) + 1; }
    void run(unsigned parts, const std::function<void(unsigned)> &fn)
    {
        if (parts == 0) return;
        if (threads_.empty() || parts == 1) {
            for (unsigned p = 0; p < parts; ++p) fn(p);
            return;
        }
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn;
            parts_ = parts;
            next_.store(0);
            left_.store(parts);
            ++job_;
        }
        cv_.notify_all();
        take();
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return left_.load() == 0 && busy_ == 0; }); // (no worker is still looking at this job when the next is posted)
        fn_ = nullptr;
    }

private:
    void take()
    {
        for (;;) {
            const unsigned p = next_.fetch_add(1);
            if (p >= parts_) return;
            (*fn_)(p);
            if (left_.fetch_sub(1) == 1) {
                std::lock_guard<std::mutex> g(m_);
                done_.notify_all();
            }
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return quit_ || job_ != seen; });
                if (quit_) return;
                seen = job_;
                if (fn_ == nullptr) continue; // (the job was over before this worker woke)
                ++busy_;
            }
            take();
            {
                std::lock_guard<std::mutex> g(m_);
                --busy_;
            }
            done_.notify_all();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(unsigned)> *fn_ = nullptr;
    unsigned parts_ = 0;
    std::atomic<unsigned> next_{0}, left_{0};
    uint64_t job_ = 0;
    unsigned busy_ = 0;
    bool quit_ = false;
};

// net/ClientConnection.java + net/JSONClientConnection.java: one peer
struct Connection {
    int fd = -1;
    std::string name;
    bool connected = false;
    int64_t emulationTime = 0; // how far this emulator has reached (setTime)
    // the framer (Framer below): which of the protocol's three kinds of input the next byte belongs to, and how far into it
    enum class Input { HeaderLine, JsonText, SizedPayload } input = Input::HeaderLine;
    bool escaped = false, inString = false; // JsonText: after a backslash / inside a string literal
    int depth = 0;                          // JsonText: open braces
    std::string text;                       // bytes of the unit being read
    int64_t payloadLeft = 0;                // SizedPayload: bytes still to come
    // pending output: `out`, and behind ALL of it the bulk segments -- a step's receive messages as the workers wrote them, one
    // string per worker and connection, sent where they lie (joining them into `out` would copy 13 MB a step at the BASELINE size)
    std::string out;
    std::deque<std::string> bulk;
    size_t bulkOff = 0;        // bytes of bulk.front() already sent
    size_t bulkBytes = 0;      // unsent bytes in the bulk segments
    std::string &tail() { return bulk.empty() ? out : bulk.back(); } // where the next message is appended
    void noteTail(size_t before) { if (!bulk.empty()) bulkBytes += bulk.back().size() - before; }
    void settleBulk() // the bulk segments' unsent rest becomes ordinary output (something has to go out BEHIND `out` but before them: never, normally)
    {
        for (size_t i = 0; i < bulk.size(); ++i) out.append(bulk[i], i == 0 ? bulkOff : 0, std::string::npos);
        bulk.clear();
        bulkOff = bulkBytes = 0;
    }
    uint64_t messagesIn = 0, messagesOut = 0;
    // The node-info array of this emulator's time-step messages as it was sent last: the nodes' objects joined by commas, kept
    // in pieces of kInfoPiece nodes (every piece but the first begins with the comma that joins it to the one before).  A
    // step rewrites only the objects of nodes whose fields changed; an object whose text changes its length -- a reception
    // beginning or ending swaps "-99.99" for seventeen digits -- moves the few KB behind it in its piece, not the 7 MB
    // behind it in the array.
    static constexpr size_t kInfoPiece = 64;
    std::vector<int32_t> infoNodes;      // this connection's nodes (node index), in registration order
    std::vector<std::string> infoPieces;
    std::vector<uint32_t> infoAt;        // offset of node k's object in its piece (piece k / kInfoPiece)
    std::vector<uint16_t> infoLen;       // ... its length, and the length of the part that never changes: {"node-id":<id>,"rssi":
    std::vector<uint16_t> infoHead;
    size_t infoBytes = 0;                // the pieces' lengths, summed
    // A time-step message in flight goes out of the pieces themselves (copying the array into `out` was most of what a step's
    // messages cost): `out` is cut at stepMark, and between the two parts go stepHead, the pieces and "]}}\r\n".  The pieces
    // are not touched while any of the message is unsent -- settleStep() copies the rest into `out` before that.
    static constexpr size_t kNoStep = ~size_t(0);
    size_t stepMark = kNoStep; // bytes of `out` that go before the step message (kNoStep: none in flight)
    size_t stepOff = 0;        // bytes of the step message already sent
    std::string stepHead;
    static const char *stepTail() { return "]}}\r\n"; }
    size_t stepLen() const { return stepHead.size() + infoBytes + 5; }
    size_t pending() const { return out.size() + (stepMark == kNoStep ? 0 : stepLen() - stepOff) + bulkBytes; }
    // the unsent part of the step message as (pointer, length) runs, at most `max_runs` of them
    template <class F> void stepRuns(F &&run, size_t max_runs) const
    {
        size_t off = stepOff, runs = 0;
        auto part = [&](const char *p, size_t len) {
            if (off >= len) {
                off -= len;
                return;
            }
            if (runs < max_runs) run(p + off, len - off);
            ++runs;
            off = 0;
        };
        part(stepHead.data(), stepHead.size());
        for (const std::string &pc : infoPieces) {
            if (runs >= max_runs) return;
            part(pc.data(), pc.size());
        }
        part(stepTail(), 5);
    }
    void settleStep() // the unsent rest of the step message becomes ordinary output
    {
        if (stepMark == kNoStep) return;
        std::string rest;
        rest.reserve(stepLen() - stepOff);
        stepRuns([&](const char *p, size_t len) { rest.append(p, len); }, ~size_t(0));
        out.insert(stepMark, rest);
        stepMark = kNoStep;
        stepOff = 0;
    }

    bool setTime(int64_t time) // JSONClientConnection.java:361-367
    {
        if (emulationTime <= time) {
            emulationTime = time;
            return true;
        }
        return false;
    }
    bool send(const Json &json) // :261-287 (useLength = false)
    {
        if (fd < 0) return false; // output == null after close()
        std::string &o = tail();
        const size_t before = o.size();
        json.append_to(o);
        o += "\r\n";
        noteTail(before);
        ++messagesOut;
        return true;
    }
    // the same for a message whose minimal JSON text the caller has written itself (the per-node and per-delivery
    // messages: no object tree for a hundred thousand node-infos)
    bool open() const { return fd >= 0; }
    void sent(size_t before) // (a message written straight into tail(), which was `before` bytes long)
    {
        tail() += "\r\n";
        noteTail(before);
        ++messagesOut;
    }
};

struct Options {
    int port = 7711;       // Simulator.DEFAULT_PORT
    int device = 0;
    bool noMedium = false;
    bool perPacket = false; // evaluate every transmit on its own instead of one evaluation per tick
    int64_t seed = 0;
    std::string pcap;
    std::string bind = "0.0.0.0";
};

class RadioLinkServer {
public:
    explicit RadioLinkServer(const Options &o) : opt_(o), sim_(o.seed)
    {
        welcome_ = Json::object();
        welcome_.set("radio-simulator", Json::object().set("name", Json::of("RSIM 0.1")).set("api-version", Json::of("0.6")));
        welcome_.set("status", Json::of("OK"));
        if (!opt_.pcap.empty()) {
            pcap_.reset(new emul8::PcapListener(opt_.pcap));
            sim_.addRadioListener(pcap_.get());
        }
        if (!opt_.noMedium) setMedium(new emul8::NullRadioMedium(opt_.device)); // Main.java:67-71
        unsigned threads = std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
        if (const char *e = std::getenv("RSIM_SERVER_THREADS")) threads = unsigned(std::max(1, std::atoi(e)));
        if (threads > 1) workers_.reset(new Workers(threads));
        if (const char *e = std::getenv("RSIM_PARALLEL_FROM")) kParallelFrom = size_t(std::max(1, std::atoi(e)));
    }

    int listenOn()
    {
        lfd_ = ::socket(AF_INET, SOCK_STREAM, 0);
        if (lfd_ < 0) throw std::runtime_error(std::string("socket: ") + std::strerror(errno));
        int one = 1;
        ::setsockopt(lfd_, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        sockaddr_in a{};
        a.sin_family = AF_INET;
        a.sin_port = htons(uint16_t(opt_.port));
        if (::inet_pton(AF_INET, opt_.bind.c_str(), &a.sin_addr) != 1) throw std::runtime_error("bad bind address " + opt_.bind);
        if (::bind(lfd_, reinterpret_cast<sockaddr *>(&a), sizeof(a)) != 0 || ::listen(lfd_, 64) != 0)
            throw std::runtime_error("Server listen on port " + std::to_string(opt_.port) + " failed: " + std::strerror(errno));
        socklen_t len = sizeof(a);
        ::getsockname(lfd_, reinterpret_cast<sockaddr *>(&a), &len);
        return ntohs(a.sin_port);
    }

    void run(volatile sig_atomic_t *stop)
    {
        std::vector<pollfd> fds;
        std::vector<Connection *> who;
        while (!*stop) {
            fds.clear();
            who.clear();
            fds.push_back({lfd_, POLLIN, 0});
            who.push_back(nullptr);
            for (auto &c : conns_)
                if (c->fd >= 0) {
                    fds.push_back({c->fd, short(POLLIN | (c->pending() == 0 ? 0 : POLLOUT)), 0});
                    who.push_back(c.get());
                }
            const int n = ::poll(fds.data(), nfds_t(fds.size()), 500);
            if (n < 0) {
                if (errno == EINTR) continue;
                throw std::runtime_error(std::string("poll: ") + std::strerror(errno));
            }
            if (fds[0].revents & POLLIN) acceptOne();
            for (size_t i = 1; i < fds.size(); ++i) {
                Connection &c = *who[i];
                if (c.fd < 0) continue;
                if (fds[i].revents & (POLLIN | POLLHUP | POLLERR)) readFrom(c);
            }
            flushAll();
            reap();
        }
        for (auto &c : conns_) close(*c);
        ::close(lfd_);
    }

    // ---- the reader: JSONClientConnection.processInput, one byte at a time (public for the framing tests)
    void feed(Connection &c, const char *data, size_t n)
    {
        try {
            for (size_t i = 0; i < n && c.connected; ++i) feedByte(c, static_cast<unsigned char>(data[i]));
        } catch (const std::exception &e) { // the reader thread's catch: log, then close()
            VLOG("%s connection closed: %s", c.name.c_str(), e.what());
            close(c);
        }
    }

private:
    // ---------------------------------------------------------------- connections
    void acceptOne()
    {
        sockaddr_in a{};
        socklen_t len = sizeof(a);
        const int fd = ::accept(lfd_, reinterpret_cast<sockaddr *>(&a), &len);
        if (fd < 0) return;
        int one = 1;
        ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
        ::fcntl(fd, F_SETFL, ::fcntl(fd, F_GETFL, 0) | O_NONBLOCK);
        char ip[64] = "?";
        ::inet_ntop(AF_INET, &a.sin_addr, ip, sizeof(ip));
        std::unique_ptr<Connection> c(new Connection);
        c->fd = fd;
        c->name = std::string("[") + ip + ":" + std::to_string(ntohs(a.sin_port)) + "]";
        c->connected = true;
        VLOG("%s client connected", c->name.c_str());
        c->send(welcome_); // Server.java:107-109
        conns_.push_back(std::move(c));
    }
    void readFrom(Connection &c)
    {
        char buf[65536];
        for (;;) {
            const ssize_t n = ::recv(c.fd, buf, sizeof(buf), 0);
            if (n > 0) {
                feed(c, buf, size_t(n));
                if (c.fd < 0) return;
                if (size_t(n) < sizeof(buf)) return;
                continue;
            }
            if (n == 0) { // read() < 0: close()
                close(c);
                return;
            }
            if (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR) return;
            close(c);
            return;
        }
    }
    void flushAll()
    {
        for (auto &cp : conns_) {
            Connection &c = *cp;
            while (c.fd >= 0 && c.pending() != 0) {
                // what is pending, in order: out[0, stepMark), the step message (head, node-info body, tail), the rest of out
                constexpr int kIov = 512;
                iovec iov[kIov + 2];
                int ni = 0;
                auto add = [&](const char *p, size_t len) {
                    if (len) iov[ni++] = iovec{const_cast<char *>(p), len};
                };
                const bool step = c.stepMark != Connection::kNoStep;
                const size_t before = step ? c.stepMark : c.out.size();
                add(c.out.data(), before);
                if (step) {
                    size_t runs_len = 0;
                    c.stepRuns([&](const char *p, size_t len) { add(p, len), runs_len += len; }, size_t(kIov));
                    // (the rest of `out` only behind the WHOLE of the message's rest: a long array takes several calls)
                    if (runs_len == c.stepLen() - c.stepOff) add(c.out.data() + before, c.out.size() - before);
                }
                // the bulk segments: only behind the WHOLE of what precedes them
                size_t ahead = 0;
                for (int k = 0; k < ni; ++k) ahead += iov[k].iov_len;
                if (ahead == c.pending() - c.bulkBytes)
                    for (size_t k = 0; k < c.bulk.size() && ni < kIov; ++k)
                        add(c.bulk[k].data() + (k == 0 ? c.bulkOff : 0), c.bulk[k].size() - (k == 0 ? c.bulkOff : 0));
                msghdr mh{};
                mh.msg_iov = iov;
                mh.msg_iovlen = size_t(ni);
                const ssize_t n = ::sendmsg(c.fd, &mh, MSG_NOSIGNAL);
                if (n > 0) {
                    size_t left = size_t(n);
                    const size_t a = std::min(left, before);
                    c.out.erase(0, a);
                    left -= a;
                    if (step) {
                        c.stepMark -= a;
                        const size_t sv = std::min(left, c.stepLen() - c.stepOff);
                        c.stepOff += sv;
                        left -= sv;
                        if (c.stepOff == c.stepLen()) {
                            c.stepMark = Connection::kNoStep;
                            c.stepOff = 0;
                        }
                        const size_t b = std::min(left, c.out.size());
                        c.out.erase(0, b);
                        left -= b;
                    }
                    while (left > 0 && !c.bulk.empty()) { // what went out of the bulk segments
                        const size_t have = c.bulk.front().size() - c.bulkOff, took = std::min(left, have);
                        c.bulkOff += took;
                        c.bulkBytes -= took;
                        left -= took;
                        if (took == have) {
                            // (the segment's room -- megabytes, its pages touched -- serves the next step's messages again)
                            if (spare_.size() < 64 && c.bulk.front().capacity() >= 4096) {
                                spare_.push_back(std::move(c.bulk.front()));
                                spare_.back().clear();
                            }
                            c.bulk.pop_front();
                            c.bulkOff = 0;
                        }
                    }
                    continue;
                }
                if (n < 0 && (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR)) break;
                close(c); // "failed to reply to client"
            }
            // a peer that has stopped reading: the reference blocks that connection's own thread; one thread serves all
            // of them here, so the connection is given up once its backlog passes the cap
            if (c.fd >= 0 && c.pending() > kMaxBacklog) {
                std::fprintf(stderr, "%s: %zu bytes unsent, peer not reading: closing\n", c.name.c_str(), c.pending());
                c.stepMark = Connection::kNoStep;
                c.out.clear();
                c.bulk.clear();
                c.bulkOff = c.bulkBytes = 0;
                close(c);
            }
        }
    }
    static constexpr size_t kMaxBacklog = size_t(256) << 20;
    void close(Connection &c) // JSONClientConnection.close: the simulator keeps whatever referred to it
    {
        if (c.fd >= 0) {
            c.settleStep();
            c.settleBulk();
            if (!c.out.empty()) { // what was already "written" in the reference's blocking send: one bounded attempt
                const timeval tv{0, 200 * 1000};   // (a peer that does not read must not stall the poll loop)
                ::setsockopt(c.fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
                ::fcntl(c.fd, F_SETFL, ::fcntl(c.fd, F_GETFL, 0) & ~O_NONBLOCK);
                (void)!::send(c.fd, c.out.data(), c.out.size(), MSG_NOSIGNAL);
                c.out.clear();
            }
            ::close(c.fd);
            VLOG("%s disconnected", c.name.c_str());
        }
        c.fd = -1;
        c.connected = false;
    }
    bool referenced(const Connection *c) const
    {
        if (c == timeController_) return true;
        if (std::find(emulators_.begin(), emulators_.end(), c) != emulators_.end()) return true;
        if (std::find(eventListeners_.begin(), eventListeners_.end(), c) != eventListeners_.end()) return true;
        return false; // a node's connection is always in emulators_
    }
    void reap() // closed connections nothing refers to any more
    {
        conns_.erase(std::remove_if(conns_.begin(), conns_.end(),
                                    [&](const std::unique_ptr<Connection> &c) { return c->fd < 0 && !referenced(c.get()); }),
                     conns_.end());
    }

    // ---------------------------------------------------------------- framing
    static void appendLatin1(std::string &s, unsigned c) // (char) c of one byte, kept as UTF-8
    {
        if (c < 0x80) s += char(c);
        else {
            s += char(0xC0 | (c >> 6));
            s += char(0x80 | (c & 0x3F));
        }
    }
    // What the wire may carry (net/JSONClientConnection.java:134-255 is the behaviour to match, byte for byte): a JSON object
    // wherever a '{' turns up -- read to its matching '}', braces inside string literals and escaped characters not counted,
    // CRs outside literals dropped, bytes taken as Latin-1 --, or a header line "<size>;attributes" followed by exactly
    // <size> bytes of UTF-8 JSON; a header without a size means "JSON follows"; blank lines and CRs between units are noise.
    // One unit complete -> one message dispatched.
    void feedByte(Connection &c, unsigned ch)
    {
        using Input = Connection::Input;
        switch (c.input) {
        case Input::SizedPayload:
            c.text += char(ch);
            if (--c.payloadLeft == 0) completeUnit(c);
            return;
        case Input::HeaderLine:
            if (ch == '{') { // an object begins, whatever the line held so far (it stays in front of the text, as in the reference)
                c.input = Input::JsonText;
                c.escaped = c.inString = false;
                break;       // ... and this brace is its first byte
            }
            if (ch == '\r') return;
            if (ch != '\n') {
                appendLatin1(c.text, ch);
                return;
            }
            headerLine(c);
            return;
        case Input::JsonText:
            break;
        }
        // JsonText
        if (ch == '\r' && !c.escaped && !c.inString) return;
        appendLatin1(c.text, ch);
        if (c.escaped) {
            c.escaped = false;
        } else if (ch == '\\') {
            c.escaped = true;
        } else if (c.inString) {
            c.inString = ch != '"';
        } else if (ch == '"') {
            c.inString = true;
        } else if (ch == '{') {
            ++c.depth;
        } else if (ch == '}' && --c.depth == 0) {
            completeUnit(c);
        }
    }
    void completeUnit(Connection &c)
    {
        std::string unit;
        unit.swap(c.text);
        c.input = Connection::Input::HeaderLine;
        dispatch(c, Json::parse_object(unit));
    }
    void headerLine(Connection &c)
    {
        std::string line;
        line.swap(c.text);
        if (line.find_first_not_of(" \t\n\v\f\r") == std::string::npos) return; // nothing but white space
        const int64_t size = parseJavaInt(line.substr(0, line.find(';')));
        if (size > 20 * 1024 * 1024) throw std::runtime_error("too large payload: " + std::to_string(size));
        if (size < 0) { // no size given: JSON text follows, to be read to its closing brace
            c.input = Connection::Input::JsonText;
            c.escaped = c.inString = false;
            c.depth = 0;
        } else if (size > 0) {
            c.input = Connection::Input::SizedPayload;
            c.payloadLeft = size;
            c.text.reserve(size_t(size));
        }
    }
    static int64_t parseJavaInt(const std::string &s) // Integer.parseInt: sign, digits, nothing else
    {
        size_t i = 0;
        if (i < s.size() && (s[i] == '-' || s[i] == '+')) ++i;
        if (i == s.size() || s.size() - i > 10) throw std::runtime_error("For input string: \"" + s + "\"");
        int64_t v = 0;
        for (size_t k = i; k < s.size(); ++k) {
            if (s[k] < '0' || s[k] > '9') throw std::runtime_error("For input string: \"" + s + "\"");
            v = v * 10 + (s[k] - '0');
        }
        if (s[0] == '-') v = -v;
        if (v > INT32_MAX || v < INT32_MIN) throw std::runtime_error("For input string: \"" + s + "\"");
        return v;
    }
    void dispatch(Connection &c, const Json &json)
    {
        ++c.messagesIn;
        VLOG("%s Got: %s", c.name.c_str(), json.toString().c_str());
        handleMessage(c, json);
    }

    // ---------------------------------------------------------------- the medium
    void setMedium(GpuRadioMedium *m)
    {
        if (medium_) medium_->flush();
        medium_.reset(m);
        packets_.clear(); // the old medium's pending events went with it (the reference's queue would still fire them)
        if (m) {
            m->setTickMode(!opt_.perPacket);
            sim_.setRadioMedium(m);
            m->setDeviceEvents(true);
        } else {
            sim_.setRadioMedium(nullptr);
        }
    }
    // queued transmissions are evaluated before anything they depend on changes (a no-op in per-packet mode)
    void settle()
    {
        if (medium_) {
            medium_->flush();
            mediumError("transmit");
        }
    }
    void mediumError(const char *where)
    {
        if (medium_ && !medium_->lastError.empty()) {
            std::fprintf(stderr, "radio medium error in %s: %s\n", where, medium_->lastError.c_str());
            medium_->lastError.clear();
        }
    }
    struct Info {
        double rssi;
        int receiving, channel;
    };
    // Transciever.getRSSI / getReceivingState / getWirelessChannel of some nodes: from the device's radio state
    std::vector<Info> nodeInfo(const std::vector<Node *> &nodes)
    {
        std::vector<Info> out(nodes.size());
        if (medium_) {
            std::vector<int32_t> idx(nodes.size()), recv, chan;
            std::vector<double> rssi;
            for (size_t i = 0; i < nodes.size(); ++i) idx[i] = nodes[i]->index;
            if (!nodes.empty() && !medium_->nodeInfo(idx, rssi, recv, chan)) {
                mediumError("node-info");
                throw std::runtime_error("node-info failed");
            }
            for (size_t i = 0; i < nodes.size(); ++i) out[i] = {rssi[i], recv[i], chan[i]};
        } else { // no medium: nothing ever starts a reception
            for (size_t i = 0; i < nodes.size(); ++i)
                out[i] = {nodes[i]->getRadio().getRSSI(), nodes[i]->getRadio().getReceivingState(), nodes[i]->getRadio().getWirelessChannel()};
        }
        return out;
    }

    // ---------------------------------------------------------------- Simulator.java's time stepping
    int64_t nextMessageId() // :118-120, :366-372
    {
        messageId_ = messageId_ == INT64_MAX ? 0 : messageId_ + 1;
        return messageId_;
    }
    Connection *connectionOf(const Node *n) const { return size_t(n->index) < nodeConn_.size() ? nodeConn_[size_t(n->index)] : nullptr; }
    Node *addNode(const std::string &id, Connection *client) // :249-277
    {
        if (Node *n = sim_.getNode(id)) return n;
        if (std::find(emulators_.begin(), emulators_.end(), client) == emulators_.end()) emulators_.push_back(client);
        Node *n = sim_.addNode(id);
        nodeConn_.resize(size_t(n->index) + 1, nullptr);
        nodeConn_[size_t(n->index)] = client;
        noteNode(n, client);
        return n;
    }
    // ---- the node-info of a time-step message (net/JSONClientConnection.java:326-353: every node of the connection, every
    // step).  Writing a hundred thousand objects per step -- a shortest-digits double each -- was 4.3 ms of a step whose
    // evaluation takes 0.8; but between two steps most nodes' fields are what they were (idle, or still locked on the same
    // frame).  So every connection keeps the array as it sent it last (Connection::infoPieces), and a step asks the device
    // which nodes differ from what was reported last (rm_node_info_changed) and rewrites those objects' fields where they lie.
    struct NodeText {
        std::string quotedId; // the node id as a JSON string
        Connection *conn = nullptr;
        uint32_t slot = 0;  // place among its connection's nodes
    };
    std::vector<NodeText> nodeText_;   // by node index
    std::vector<int32_t> chgIdx_, chgRecv_, chgChan_;
    std::vector<double> chgRssi_;
    // the part of a node's object behind "rssi": -- <rssi>,"receiving":<r>,"wireless-channel":<c>} -- into buf (at least 112 bytes)
    static size_t writeNodeFields(char *buf, const Info &v)
    {
        std::string tmp; // (short: stays in the string's own buffer for every value but a seventeen-digit rssi)
        append_double(tmp, v.rssi);
        size_t n = tmp.size();
        std::memcpy(buf, tmp.data(), n);
        auto lit = [&](const char *t) {
            const size_t l = std::strlen(t);
            std::memcpy(buf + n, t, l);
            n += l;
        };
        auto num = [&](int64_t x) { n = size_t(std::to_chars(buf + n, buf + n + 24, x).ptr - buf); };
        lit(",\"receiving\":");
        num(v.receiving);
        lit(",\"wireless-channel\":");
        num(v.channel);
        buf[n++] = '}';
        return n;
    }
    void noteNode(Node *n, Connection *c) // a node seen for the first time: its object joins its connection's array
    {
        if (nodeText_.size() <= size_t(n->index)) nodeText_.resize(size_t(n->index) + 1);
        NodeText &t = nodeText_[size_t(n->index)];
        t.quotedId.clear();
        Json::quote(n->getId(), t.quotedId);
        t.conn = c;
        t.slot = uint32_t(c->infoNodes.size());
        c->settleStep(); // (a step message in flight reads the pieces)
        c->infoNodes.push_back(n->index);
        if (t.slot % Connection::kInfoPiece == 0) c->infoPieces.emplace_back();
        std::string &piece = c->infoPieces.back();
        const size_t size0 = piece.size();
        if (t.slot) piece += ',';
        const size_t at = piece.size();
        piece += "{\"node-id\":";
        piece += t.quotedId;
        piece += ",\"rssi\":";
        const size_t head = piece.size() - at;
        char buf[128];
        piece.append(buf, writeNodeFields(buf, {n->getRadio().getRSSI(), n->getRadio().getReceivingState(), n->getRadio().getWirelessChannel()}));
        if (piece.size() - at > 65000) throw std::runtime_error("node id too long for a node-info object");
        c->infoAt.push_back(uint32_t(at));
        c->infoHead.push_back(uint16_t(head));
        c->infoLen.push_back(uint16_t(piece.size() - at));
        c->infoBytes += piece.size() - size0;
    }
    // rewrites the node's fields where they lie in its piece; returns by how much the piece's length changed (the caller keeps the
    // connection's infoBytes: pieces are rewritten by several workers at once, each piece by one of them)
    long applyNodeInfo(int32_t index, const Info &v)
    {
        const NodeText &t = nodeText_[size_t(index)];
        Connection &c = *t.conn;
        char buf[128];
        const size_t len = writeNodeFields(buf, v);
        std::string &piece = c.infoPieces[t.slot / Connection::kInfoPiece];
        const size_t at = size_t(c.infoAt[t.slot]) + c.infoHead[t.slot], before = size_t(c.infoLen[t.slot]) - c.infoHead[t.slot];
        if (len == before) { // same length: in place
            std::memcpy(&piece[at], buf, len);
            return 0;
        }
        piece.replace(at, before, buf, len);
        const size_t last = std::min(c.infoNodes.size(), (t.slot / Connection::kInfoPiece + 1) * Connection::kInfoPiece);
        for (size_t k = size_t(t.slot) + 1; k < last; ++k) c.infoAt[k] = uint32_t(c.infoAt[k] + len - before);
        c.infoLen[t.slot] = uint16_t(c.infoHead[t.slot] + len);
        return long(len) - long(before);
    }
    // the changed nodes' objects, rewritten by the workers: a piece (64 objects) belongs to ONE worker, so the changes are first
    // dealt to their pieces' owners (every worker deals its share of the list), then every owner rewrites what it was dealt
    void applyNodeInfoParallel(const std::vector<int32_t> &idx, const std::vector<double> &rssi, const std::vector<int32_t> &recv,
                               const std::vector<int32_t> &chan)
    {
        const unsigned T = workers_->size();
        const size_t n = idx.size();
        dealt_.resize(size_t(T) * T);
        for (auto &d : dealt_) d.clear();
        workers_->run(T, [&](unsigned t) {
            const size_t k0 = n * t / T, k1 = n * (t + 1) / T;
            for (size_t k = k0; k < k1; ++k) {
                if (size_t(idx[k]) >= nodeText_.size()) continue;
                const NodeText &nt = nodeText_[size_t(idx[k])];
                if (!nt.conn) continue;
                const size_t piece = nt.slot / Connection::kInfoPiece;
                const unsigned owner = unsigned((piece + (reinterpret_cast<uintptr_t>(nt.conn) >> 6)) % T);
                dealt_[size_t(t) * T + owner].push_back(uint32_t(k));
            }
        });
        std::vector<std::vector<std::pair<Connection *, long>>> grown(T);
        workers_->run(T, [&](unsigned owner) {
            auto &mine = grown[owner];
            for (unsigned from = 0; from < T; ++from)
                for (uint32_t k : dealt_[size_t(from) * T + owner]) {
                    const long d = applyNodeInfo(idx[k], {rssi[k], recv[k], chan[k]});
                    if (d == 0) continue;
                    Connection *c = nodeText_[size_t(idx[k])].conn;
                    if (mine.empty() || mine.back().first != c) mine.emplace_back(c, 0L);
                    mine.back().second += d;
                }
        });
        for (const auto &g : grown)
            for (const auto &cd : g) cd.first->infoBytes = size_t(long(cd.first->infoBytes) + cd.second);
    }
    // once per step, before the time-step messages: bring the nodes' texts up to the device's radio state
    void refreshNodeInfo()
    {
        if (medium_) {
            std::vector<int32_t> &idx = chgIdx_, &recv = chgRecv_, &chan = chgChan_; // (members: their room is kept between the steps)
            std::vector<double> &rssi = chgRssi_;
            if (!medium_->nodeInfoChanged(idx, rssi, recv, chan)) {
                mediumError("node-info");
                throw std::runtime_error("node-info failed");
            }
            if (workers_ && idx.size() >= kParallelFrom) {
                applyNodeInfoParallel(idx, rssi, recv, chan);
            } else {
                for (size_t k = 0; k < idx.size(); ++k)
                    if (size_t(idx[k]) < nodeText_.size() && nodeText_[size_t(idx[k])].conn) {
                        Connection *c = nodeText_[size_t(idx[k])].conn;
                        c->infoBytes = size_t(long(c->infoBytes) + applyNodeInfo(idx[k], {rssi[k], recv[k], chan[k]}));
                    }
            }
            nodeInfoChanges_ += idx.size();
            if (stepMessages_ > 0) nodeInfoChangesLater_ += idx.size();
        } else { // no medium: nothing ever starts a reception, the host's radios are the state
            for (Node *n : sim_.getNodes())
                if (size_t(n->index) < nodeText_.size() && nodeText_[size_t(n->index)].conn) {
                    Connection *c = nodeText_[size_t(n->index)].conn;
                    c->infoBytes = size_t(long(c->infoBytes) + applyNodeInfo(n->index, {n->getRadio().getRSSI(), n->getRadio().getReceivingState(),
                                                                                            n->getRadio().getWirelessChannel()}));
                }
        }
    }
    void emulateToTime(Connection &c, int64_t time, int64_t timeId) // JSONClientConnection.java:326-353
    {
        if (!c.open()) return; // send() on a closed connection: nothing goes out
        // {"command":"time-step","id":..,"parameters":{"time":..,"node-info":[{..},..]}} -- written directly: this is
        // the one message whose size grows with the node count, and its array goes out of the pieces themselves (Connection::stepMark)
        std::string &o = c.stepHead;
        o.assign("{\"command\":\"time-step\",\"id\":");
        append_int(o, timeId);
        o += ",\"parameters\":{\"time\":";
        append_int(o, time);
        o += ",\"node-info\":[";
        c.settleBulk(); // (receive messages of the step before that the peer has not taken yet go out first)
        c.stepMark = c.out.size();
        c.stepOff = 0;
        ++c.messagesOut;
    }
    void stepTime(int64_t time, int64_t id) // :171-194
    {
        if (emulatorsLeft_ > 0) VLOG("*** still waiting for %d clients when stepping time again to %lld", emulatorsLeft_, (long long)time);
        waitingForTimeId_ = nextMessageId();
        timeControllerLastTimeId_ = id;
        stepTime_ = time;
        if (emulators_.empty()) {
            emulatorsLeft_ = 0;
            emulatorTimeStepDone();
            return;
        }
        emulatorsLeft_ = int(emulators_.size());
        const std::vector<Connection *> em = emulators_;
        const auto t0 = std::chrono::steady_clock::now();
        for (Connection *e : em) e->settleStep(); // (a step message still in flight: its unsent rest is copied before the body changes)
        refreshNodeInfo();
        for (Connection *e : em) emulateToTime(*e, time, waitingForTimeId_);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (stepMessages_++ == 0) usFirstStepMessages_ = us; // (the first one writes every node's object: reported on its own)
        else usStepMessages_ += us;
    }
    void emulatorTimeStepped(Connection &client, int64_t id) // :134-153
    {
        if (waitingForTimeId_ < 0) return;
        if (id != waitingForTimeId_) return;
        if (client.setTime(stepTime_)) emulatorsLeft_--;
        if (emulatorsLeft_ == 0) emulatorTimeStepDone();
    }
    void emulatorTimeStepDone() // :155-165
    {
        waitingForTimeId_ = -1;
        // queued transmissions are evaluated at the old time, the clock moves, the events up to the new time fire:
        // the deliveries come back in the reference queue's pop order
        const auto t0 = std::chrono::steady_clock::now();
        sim_.calls.clear();
        sim_.emulatorTimeStepDone(stepTime_);
        mediumError("time step");
        const auto t1 = std::chrono::steady_clock::now();
        framed_ = Framed();
        if (workers_ && sim_.calls.size() >= kParallelFrom) {
            deliverParallel(sim_.calls);
        } else {
            for (const emul8::MediumCall &call : sim_.calls)
                if (call.kind == emul8::MediumCall::DELIVER) deliverRadioPacket(*call.packet, *call.destination, call.rssi);
        }
        sim_.calls.clear();
        usMedium_ += std::chrono::duration<double, std::micro>(t1 - t0).count();
        usReceiveMessages_ += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
        ++steps_;
        prunePackets();
        if (timeController_) timeController_->send(Json::object().add("reply", Json::of("OK")).add("id", Json::of(timeControllerLastTimeId_)));
    }
    // RadioPacket.toJsonDestination, written directly (tens of thousands per tick).  The deliveries of one packet come one after
    // the other, and all but the receiver and its rssi is the packet's: that text is put together once per packet; the rssi's
    // digits are kept while it stays the same (the reference's media hand the packet's transmit power to every receiver).
    struct Framed {
        const RadioPacket *packet = nullptr;
        std::string head, tail, rssiText;
        double rssi = 0;
        bool haveRssi = false;
    };
    void writeReceive(std::string &o, const RadioPacket &p, const Node &dst, double rssi, Framed &f) const
    {
        if (&p != f.packet) {
            f.packet = &p;
            f.head.assign(",\"time-start\":");
            append_int(f.head, p.getStartTime());
            f.head += ",\"time-end\":";
            append_int(f.head, p.getEndTime());
            f.head += ",\"rf-power\":";
            f.tail.assign(",\"wireless-channel\":");
            append_int(f.tail, p.getWirelessChannel());
            f.tail += ",\"packet-data\":";
            Json::quote(p.getPacketDataAsHex(), f.tail);
            f.tail += "}\r\n";
        }
        if (!f.haveRssi || std::memcmp(&rssi, &f.rssi, sizeof(double)) != 0) {
            f.rssi = rssi;
            f.haveRssi = true;
            f.rssiText.clear();
            append_double(f.rssiText, rssi);
        }
        o += "{\"command\":\"receive\",\"node-id\":";
        if (size_t(dst.index) < nodeText_.size() && !nodeText_[size_t(dst.index)].quotedId.empty()) o += nodeText_[size_t(dst.index)].quotedId;
        else Json::quote(dst.getId(), o);
        o += f.head;
        o += f.rssiText;
        o += f.tail;
    }
    void deliverRadioPacket(const RadioPacket &p, Node &dst, double rssi) // :356-364 + RadioPacket.toJsonDestination
    {
        Connection *cc = connectionOf(&dst);
        if (!cc || !cc->connected) {
            VLOG("Node %s has no client connection", dst.getId().c_str());
            return;
        }
        std::string &o = cc->tail();
        const size_t before = o.size();
        writeReceive(o, p, dst, rssi, framed_);
        cc->noteTail(before);
        ++cc->messagesOut;
        ++deliveries_;
    }
    // a step's deliveries written by the workers: contiguous shares of the call list, each into its own string per connection;
    // the strings join their connections' output as segments, in the order of the shares -- a connection's messages stay in the
    // queue's pop order, and nothing is copied a second time
    void deliverParallel(const std::vector<emul8::MediumCall> &calls)
    {
        const unsigned T = workers_->size();
        struct Share {
            std::vector<std::pair<Connection *, std::string>> per;
            std::vector<uint64_t> count;
            std::string spare;
        };
        std::vector<Share> shares(T);
        const size_t n = calls.size();
        for (unsigned t = 0; t < T && !spare_.empty(); ++t) { // a string that has held a share's messages before, for each share's first connection
            shares[t].spare = std::move(spare_.back());
            spare_.pop_back();
        }
        workers_->run(T, [&](unsigned t) {
            Share &sh = shares[t];
            Framed f;
            Connection *last = nullptr;
            size_t lastAt = 0;
            for (size_t k = n * t / T; k < n * (t + 1) / T; ++k) {
                const emul8::MediumCall &call = calls[k];
                if (call.kind != emul8::MediumCall::DELIVER) continue;
                Connection *cc = connectionOf(call.destination);
                if (!cc || !cc->connected) continue;
                if (cc != last) {
                    last = cc;
                    for (lastAt = 0; lastAt < sh.per.size() && sh.per[lastAt].first != cc; ++lastAt) {}
                    if (lastAt == sh.per.size()) {
                        sh.per.emplace_back(cc, std::string());
                        if (sh.per.size() == 1 && sh.spare.capacity() != 0) sh.per.back().second = std::move(sh.spare);
                        else sh.per.back().second.reserve(size_t(420) * (n / T + 1) / std::max<size_t>(1, sh.per.size()));
                        sh.count.push_back(0);
                    }
                }
                writeReceive(sh.per[lastAt].second, *call.packet, *call.destination, call.rssi, f);
                ++sh.count[lastAt];
            }
        });
        for (Share &sh : shares)
            for (size_t i = 0; i < sh.per.size(); ++i) {
                Connection *cc = sh.per[i].first;
                if (sh.per[i].second.empty()) continue;
                cc->bulkBytes += sh.per[i].second.size();
                cc->bulk.push_back(std::move(sh.per[i].second));
                cc->messagesOut += sh.count[i];
                deliveries_ += sh.count[i];
            }
    }
    // The medium refers to a packet until its last event has fired, or until a failed evaluation dropped it; it
    // names the packets it has let go of (by identity -- counting from the front of a queue would free an older,
    // still pending packet in place of a newer one a failed flush never took).
    void prunePackets()
    {
        if (!medium_) {
            packets_.clear();
            return;
        }
        released_.clear();
        medium_->takeReleased(released_);
        for (RadioPacket *p : released_) packets_.erase(p);
    }

    // ---------------------------------------------------------------- SimulatorJSONHandler.handleMessage
    static Json replyObject(int64_t id) // :256-263
    {
        Json r = Json::object();
        if (id >= 0) r.set("id", Json::of(id));
        r.set("reply", Json::of("OK"));
        return r;
    }
    static Json replyError(int64_t id, const std::string &cls, const std::string &description) // :265-273
    {
        Json r = Json::object();
        if (id >= 0) r.set("id", Json::of(id));
        r.set("reply", Json::of("error"));
        r.set("reply-object", Json::object().add("class", Json::of(cls)).add("description", Json::of(description)));
        return r;
    }
    static bool isNumber(const Json *v) { return v && v->isNumber(); }

    // ---- one message in, at most one reply out.
    // A message is either an emulator's answer to a time step ("reply") or a request ("command").  Requests go through a
    // table: name -> handler, and what the sender is owed afterwards.  A handler answers by itself (Answer::sent), asks for
    // the protocol's plain acknowledgement ({"id":..,"reply":"OK"}, only if the request carried an id >= 0), or -- time-set --
    // leaves the answer to the moment the step completes.  The wire behaviour (strings, member order, which malformed input
    // ends the connection and which gets an error reply) is what net/SimulatorJSONHandler.java:28-274 does; it is pinned by
    // tests/test_host_server.py and tests/test_gpu_server.py, not by the shape of this code.
    enum class Answer { ack, sent, later };
    struct Request {
        Connection &from;
        const Json &body;
        int64_t id; // -1: no reply wanted
    };
    using Handler = Answer (RadioLinkServer::*)(const Request &);
    struct Command {
        const char *name;
        Handler run;
    };
    static const Command *findCommand(const std::string &name)
    {
        static const Command table[] = {
            {"time-get", &RadioLinkServer::onTimeGet},
            {"time-set", &RadioLinkServer::onTimeSet},
            {"transmit", &RadioLinkServer::onTransmit},
            {"log", &RadioLinkServer::onLog},
            {"node-config-set", &RadioLinkServer::onNodeConfig},
            {"link-quality", &RadioLinkServer::onLinkQuality},
            {"configuration-set", &RadioLinkServer::onConfiguration},
            {"subscribe-event", &RadioLinkServer::onSubscribe},
            {"unsubscribe-event", &RadioLinkServer::onUnsubscribe},
        };
        for (const Command &c : table)
            if (name == c.name) return &c;
        return nullptr;
    }
    void refuse(const Request &rq, const std::string &why) { rq.from.send(replyError(rq.id, "command-error", why)); }

    void handleMessage(Connection &client, const Json &json)
    {
        if (json.get("reply")) { // an emulator answering a time-step (a non-string "reply" throws: the connection ends)
            const std::string status = json.at("reply").asString();
            const int64_t step = json.getLong("id", -1);
            if (status != "OK") VLOG("%s error reply: %s", client.name.c_str(), json.toString().c_str());
            else if (step >= 0 && step == waitingForTimeId_) emulatorTimeStepped(client, step);
            return;
        }
        const Request rq{client, json, json.getLong("id", -1)};
        const Json *name = json.get("command");
        if (!name) return refuse(rq, "no command specified");
        const std::string word = name->asString();
        const Command *cmd = findCommand(word);
        if (!cmd) return refuse(rq, "unsupported command: " + word);
        if ((this->*cmd->run)(rq) == Answer::ack && rq.id >= 0) client.send(replyObject(rq.id));
    }

    Answer onTimeGet(const Request &rq)
    {
        if (rq.id < 0) return Answer::sent; // nobody asked
        rq.from.send(replyObject(rq.id).set("reply-object", Json::object().add("time", Json::of(sim_.getTime()))));
        return Answer::sent;
    }
    // the first connection to set the time is the time controller from then on; its reply comes when every emulator has stepped
    Answer onTimeSet(const Request &rq)
    {
        if (rq.id < 0) return refuse(rq, "time-set must include reply id"), Answer::sent;
        if (!timeController_) timeController_ = &rq.from;
        if (timeController_ != &rq.from) return refuse(rq, "only one time controller allowed"), Answer::sent;
        try {
            stepTime(member(member(rq.body, "parameters").asObject(), "time").asLong(), rq.id);
            return Answer::later;
        } catch (const MissingMember &) { // (the reference reports a NullPointerException's null message here)
            refuse(rq, "failed to set time:null");
        } catch (const std::exception &e) {
            refuse(rq, std::string("failed to set time:") + e.what());
        }
        return Answer::sent;
    }
    Answer onTransmit(const Request &rq)
    {
        const std::string who = member(rq.body, "node-id").toString();
        const int64_t when = member(rq.body, "time").asLong();
        const std::string payloadHex = rq.body.getString("packet-data", "");
        Node *sender = sim_.getNode(who);
        if (!sender) {
            VLOG("non-existing node sending radio packet: %s", who.c_str());
            return refuse(rq, "could not find source node"), Answer::sent;
        }
        if (!medium_) return refuse(rq, "no radio medium available"), Answer::sent;
        RadioPacket *frame = new RadioPacket(sender, when, payloadHex);
        packets_.emplace(frame, std::unique_ptr<RadioPacket>(frame));
        if (const Json *v = rq.body.get("rf-power"); isNumber(v)) frame->setTransmitPower(v->asDouble());
        if (const Json *v = rq.body.get("wireless-channel"); isNumber(v)) frame->setWirelessChannel(v->asInt());
        sim_.notifyRadioListeners(*frame);
        medium_->transmit(*frame);
        mediumError("transmit");
        ++transmissions_;
        return Answer::ack;
    }
    Answer onLog(const Request &rq)
    {
        const Json &args = member(rq.body, "parameters").asObject();
        const std::string who = member(args, "node-id").toString();
        const std::string text = member(args, "message").asString();
        Node *source = sim_.getNode(who);
        if (!source) throw std::runtime_error("log from a node that does not exist: " + who); // (ends the connection, as the reference's null dereference does)
        deliverLogEvent(*source, text);
        return Answer::ack;
    }
    // creates the node on first sight; every attribute is optional and applied only if it has the right JSON type
    Answer onNodeConfig(const Request &rq)
    {
        const Json &args = member(rq.body, "parameters").asObject();
        const std::string who = member(args, "node-id").toString();
        settle(); // what was sent before this message saw the nodes as they were
        Node *node = addNode(who, &rq.from);
        if (const Json *at = args.get("position"); at && at->isArray() && at->size() > 1) {
            // (deviation: Double.parseDouble("1e999") is Infinity and the reference would keep it, leaving the node unheard;
            // the device-resident table takes finite coordinates only, so such a position is not applied)
            const double px = (*at)[0].asDouble(), py = (*at)[1].asDouble(), pz = at->size() > 2 ? (*at)[2].asDouble() : 0.0;
            if (!std::isfinite(px) || !std::isfinite(py) || !std::isfinite(pz)) std::fprintf(stderr, "node %s: non-finite position ignored\n", who.c_str());
            else if (at->size() > 2) node->getPosition().set(px, py, pz);
            else node->getPosition().set(px, py);
        }
        sim_.nodeChanged(node);
        emul8::Transciever &radio = node->getRadio();
        struct {
            const char *key;
            void (*apply)(emul8::Transciever &, const Json &);
        } static const numeric[] = {
            {"rf-power", [](emul8::Transciever &r, const Json &v) { r.setTransmitPower(v.asDouble()); }},
            {"wireless-channel", [](emul8::Transciever &r, const Json &v) { r.setWirelessChannel(v.asInt()); }},
            {"rx-loss", [](emul8::Transciever &r, const Json &v) { r.setRxProbability(v.asDouble()); }},
            {"tx-loss", [](emul8::Transciever &r, const Json &v) { r.setTxProbability(v.asDouble()); }},
        };
        for (const auto &attr : numeric)
            if (const Json *v = args.get(attr.key); isNumber(v)) attr.apply(radio, *v);
        if (const Json *v = args.get("radio-state"); v && v->isString()) radio.setEnabled(v->asString() != "disabled");
        if (rq.id < 0) return Answer::sent;
        const Info now = nodeInfo({node})[0];
        Json described = Json::object();
        described.add("node-id", Json::of(who)).add("rssi", Json::of(now.rssi)).add("receiving", Json::of(now.receiving));
        described.add("wireless-channel", Json::of(now.channel));
        rq.from.send(replyObject(rq.id).set("reply-object", Json::object().add("node-info", described)));
        return Answer::sent;
    }
    // parsed for its types (a malformed one ends the connection), then dropped: the reference never applies link qualities
    Answer onLinkQuality(const Request &rq)
    {
        const Json &link = member(rq.body, "link").asObject();
        (void)member(link, "src").toString();
        (void)member(link, "dst").toString();
        if (const Json *v = rq.body.get("wireless-channel"); isNumber(v)) (void)v->asInt();
        if (const Json *v = link.get("quality"); isNumber(v)) (void)v->asInt();
        return Answer::ack;
    }
    // the medium is chosen by "propagation-option", before the first time-set only
    Answer onConfiguration(const Request &rq)
    {
        if (timeController_) return refuse(rq, "already initialized"), Answer::sent;
        const Json &args = member(rq.body, "parameters").asObject();
        const Json *chosen = args.get("propagation-option");
        if (!chosen) return Answer::ack;
        const std::string option = chosen->asString();
        if (option == "nullrm") return Answer::ack; // the default medium
        if (option == "n2n-link") return configureMatrix(rq, args);
        if (option == "log-distance") return configureLogDistance(rq, args);
        if (option == "udgm") {
            if (!opt_.noMedium) setMedium(new emul8::UDGMRadioMedium(opt_.device));
        } else if (option == "udgm-constant-loss") { // (the reference has the class but no option for it)
            if (!opt_.noMedium) setMedium(new emul8::UDGMConstantLossRadioMedium(opt_.device));
        } else {
            std::fprintf(stderr, "Unsupported propagation-option: %s - reverting to null radio medium\n", option.c_str());
        }
        return Answer::ack;
    }
    Answer configureMatrix(const Request &rq, const Json &args)
    {
        const Json *cells = args.get("matrix-data");
        if (!cells || !cells->isArray() || cells->size() == 0) return refuse(rq, "no matrix specified"), Answer::sent;
        const int n = member(args, "number-of-nodes").asInt();
        if (n != int(std::sqrt(double(cells->size())))) return refuse(rq, "inconsistent data matrix or nodes"), Answer::sent;
        std::vector<std::vector<double>> rows(size_t(n), std::vector<double>(size_t(n), 0.0));
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) rows[size_t(i)][size_t(j)] = (*cells)[size_t(i * n + j)].asDouble();
        if (!opt_.noMedium) setMedium(new emul8::N2NRadioMedium(rows, opt_.device));
        return Answer::ack;
    }
    // the engine's extension medium: the only additions to the wire are this option string and its
    // optional numeric parameters
    Answer configureLogDistance(const Request &rq, const Json &args)
    {
        rm_model_params p;
        rm_model_defaults(&p, RM_MODEL_LOGDIST);
        struct {
            const char *key;
            double rm_model_params::*field;
        } static const knobs[] = {
            {"reference-loss-db", &rm_model_params::ld_pl0_db},     {"path-loss-exponent", &rm_model_params::ld_exponent},
            {"reference-distance", &rm_model_params::ld_d0},         {"shadowing-sigma-db", &rm_model_params::ld_sigma_db},
            {"shadowing-clip", &rm_model_params::ld_clip},           {"sensitivity-dbm", &rm_model_params::ld_sensitivity_dbm},
            {"noise-dbm", &rm_model_params::ld_noise_dbm},           {"capture-db", &rm_model_params::ld_capture_db},
            {"interference-floor-dbm", &rm_model_params::ld_ifloor_dbm},
        };
        for (const auto &k : knobs)
            if (const Json *v = args.get(k.key); isNumber(v)) p.*(k.field) = v->asDouble();
        if (const Json *v = args.get("shadowing-seed"); isNumber(v)) p.ld_seed = uint64_t(v->asLong());
        if (const Json *v = args.get("sinr"); v && v->type() == Json::BOOL && v->toString() == "true") p.flags |= RM_LD_SINR;
        if (opt_.noMedium) return Answer::ack;
        std::unique_ptr<emul8::LogDistanceRadioMedium> m(new emul8::LogDistanceRadioMedium(opt_.device));
        m->params() = p;
        try {
            m->apply();
        } catch (const std::exception &e) {
            return refuse(rq, std::string("log-distance: ") + e.what()), Answer::sent;
        }
        setMedium(m.release());
        return Answer::ack;
    }
    Answer onSubscribe(const Request &rq)
    {
        eventListeners_.push_back(&rq.from); // (a second subscription means a second copy of every event, as in the reference)
        return Answer::ack;
    }
    Answer onUnsubscribe(const Request &rq)
    {
        auto it = std::find(eventListeners_.begin(), eventListeners_.end(), &rq.from); // one subscription per request
        if (it != eventListeners_.end()) eventListeners_.erase(it);
        return Answer::ack;
    }
    struct MissingMember : JsonError {
        explicit MissingMember(const std::string &n) : JsonError("missing member \"" + n + "\"") {}
    };
    static const Json &member(const Json &o, const std::string &name)
    {
        const Json *v = o.get(name);
        if (!v) throw MissingMember(name);
        return *v;
    }
    void deliverLogEvent(Node &source, const std::string &logMsg) // Node.log + JSONClientConnection.sendEvent
    {
        Json eventObject = Json::object();
        eventObject.add("time", Json::of(sim_.getTime()));
        eventObject.add("type", Json::of("log"));
        eventObject.add("source", Json::of(source.getId()));
        eventObject.add("event-data", Json::object().add("logMessage", Json::of(logMsg)));
        Json json = Json::object();
        json.add("event", eventObject);
        json.add("id", Json::of(0));
        const std::vector<Connection *> listeners = eventListeners_;
        for (Connection *l : listeners) l->send(json);
    }

public:
    void printStats() const
    {
        const double k = steps_ ? 1.0 / double(steps_) : 0.0;
        const double k1 = stepMessages_ > 1 ? 1.0 / double(stepMessages_ - 1) : 0.0;
        std::fprintf(stderr,
                     "rsim_server: %llu steps, %llu transmissions, %llu deliveries, time %lld; per step: time-step messages %.1f us "
                     "(the first one, with every node's object written, %.1f us; %.1f node objects rewritten per step after it), "
                     "medium (tick + drain, deliveries on the host) %.1f us, receive messages %.1f us\n",
                     (unsigned long long)steps_, (unsigned long long)transmissions_, (unsigned long long)deliveries_,
                     (long long)sim_.getTime(), usStepMessages_ * k1, usFirstStepMessages_, double(nodeInfoChangesLater_) * k1, usMedium_ * k,
                     usReceiveMessages_ * k);
    }

private:
    Options opt_;
    emul8::Simulator sim_;
    std::unique_ptr<GpuRadioMedium> medium_;
    std::unique_ptr<emul8::PcapListener> pcap_;
    Json welcome_;
    int lfd_ = -1;
    std::vector<std::unique_ptr<Connection>> conns_;
    std::vector<Connection *> nodeConn_; // Node.getClientConnection, by node index
    std::unordered_map<RadioPacket *, std::unique_ptr<RadioPacket>> packets_; // owned until the medium releases them
    std::vector<RadioPacket *> released_;
    // Simulator.java:69-78
    Connection *timeController_ = nullptr;
    std::vector<Connection *> emulators_, eventListeners_;
    int emulatorsLeft_ = 0;
    int64_t stepTime_ = 0, timeControllerLastTimeId_ = -1, waitingForTimeId_ = -1, messageId_ = 1000;
    uint64_t steps_ = 0, transmissions_ = 0, deliveries_ = 0;
    double usStepMessages_ = 0, usMedium_ = 0, usReceiveMessages_ = 0; // the server's own work per step (printStats)
    Framed framed_; // deliverRadioPacket: the packet whose constant text is at hand
    std::vector<std::string> spare_; // bulk segments that have been sent: their room for the next step's receive messages
    // the workers (Workers above): RSIM_SERVER_THREADS, default min(hardware threads, 8); 1 = everything on the protocol's thread
    std::unique_ptr<Workers> workers_;
    size_t kParallelFrom = 2048; // changed nodes / medium calls of a step from which the workers are worth waking (RSIM_PARALLEL_FROM: tests)
    std::vector<std::vector<uint32_t>> dealt_;    // applyNodeInfoParallel: [dealer][owner] -> places in the change list
    uint64_t stepMessages_ = 0, nodeInfoChangesLater_ = 0;
    double usFirstStepMessages_ = 0;
    uint64_t nodeInfoChanges_ = 0;     // node objects rewritten for time-step messages (the rest went out as they were)
};

} // namespace rsim

static volatile sig_atomic_t g_stop = 0;
static void onSignal(int) { g_stop = 1; }

static void usage(int code)
{
    std::puts("Usage: rsim_server [-pcap [file]] [--port N] [--bind ADDR] [--device N] [--seed N] [--per-packet] [--no-medium] [--verbose]");
    std::exit(code);
}

int main(int argc, char **argv)
{
    rsim::Options opt;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto value = [&]() -> const char * {
            if (i + 1 >= argc) usage(1);
            return argv[++i];
        };
        if (a == "-pcap") {
            if (i + 1 < argc && argv[i + 1][0] != '-') opt.pcap = argv[++i];
            else opt.pcap = "radiolog-" + std::to_string(std::chrono::duration_cast<std::chrono::milliseconds>(
                                                             std::chrono::system_clock::now().time_since_epoch()).count()) + ".pcap";
        } else if (a == "--port") opt.port = std::atoi(value());
        else if (a == "--bind") opt.bind = value();
        else if (a == "--device") opt.device = std::atoi(value());
        else if (a == "--seed") opt.seed = std::atoll(value());
        else if (a == "--per-packet") opt.perPacket = true;
        else if (a == "--no-medium") opt.noMedium = true;
        else if (a == "--verbose") rsim::g_verbose = true;
        else if (a == "-h" || a == "--help") usage(0);
        else {
            std::fprintf(stderr, "Unhandled argument: %s\n", a.c_str());
            usage(1);
        }
    }
    std::signal(SIGINT, onSignal);
    std::signal(SIGTERM, onSignal);
    std::signal(SIGPIPE, SIG_IGN);
    try {
        rsim::RadioLinkServer server(opt);
        const int port = server.listenOn();
        std::printf("Server started. Waiting for client connections at port %d.\n", port);
        std::fflush(stdout);
        server.run(&g_stop);
        server.printStats();
    } catch (const std::exception &e) {
        std::fprintf(stderr, "rsim_server: %s\n", e.what());
        return 1;
    }
    return 0;
}
