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
#include <unistd.h>

#include <algorithm>
#include <cmath>
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

// net/ClientConnection.java + net/JSONClientConnection.java: one peer
struct Connection {
    int fd = -1;
    std::string name;
    bool connected = false;
    int64_t emulationTime = 0; // how far this emulator has reached (setTime)
    // processInput's state
    bool parsingJson = false, stuffed = false, quoted = false;
    int brackets = 0;
    std::string sb;
    int64_t rawLeft = 0; // bytes of a length-prefixed payload still to read
    std::string raw;
    // pending output
    std::string out;
    uint64_t messagesIn = 0, messagesOut = 0;

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
        json.append_to(out);
        out += "\r\n";
        ++messagesOut;
        return true;
    }
    // the same for a message whose minimal JSON text the caller has written itself (the per-node and per-delivery
    // messages: no object tree for a hundred thousand node-infos)
    bool open() const { return fd >= 0; }
    void sent()
    {
        out += "\r\n";
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
                    fds.push_back({c->fd, short(POLLIN | (c->out.empty() ? 0 : POLLOUT)), 0});
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
            while (c.fd >= 0 && !c.out.empty()) {
                const ssize_t n = ::send(c.fd, c.out.data(), c.out.size(), MSG_NOSIGNAL);
                if (n > 0) {
                    c.out.erase(0, size_t(n));
                    continue;
                }
                if (n < 0 && (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR)) break;
                close(c); // "failed to reply to client"
            }
            // a peer that has stopped reading: the reference blocks that connection's own thread; one thread serves all
            // of them here, so the connection is given up once its backlog passes the cap
            if (c.fd >= 0 && c.out.size() > kMaxBacklog) {
                std::fprintf(stderr, "%s: %zu bytes unsent, peer not reading: closing\n", c.name.c_str(), c.out.size());
                c.out.clear();
                close(c);
            }
        }
    }
    static constexpr size_t kMaxBacklog = size_t(256) << 20;
    void close(Connection &c) // JSONClientConnection.close: the simulator keeps whatever referred to it
    {
        if (c.fd >= 0) {
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
    void feedByte(Connection &c, unsigned ch)
    {
        if (c.rawLeft > 0) { // a length-prefixed payload: bytes as they are, UTF-8
            c.raw += char(ch);
            if (--c.rawLeft == 0) {
                std::string text;
                text.swap(c.raw);
                dispatch(c, Json::parse_object(text));
            }
            return;
        }
        if (ch == '{') c.parsingJson = true;
        if (c.parsingJson) {
            if (ch == '\r' && !c.stuffed && !c.quoted) return;
            appendLatin1(c.sb, ch);
            if (c.stuffed) c.stuffed = false;
            else if (ch == '\\') c.stuffed = true;
            else if (c.quoted) {
                if (ch == '"') c.quoted = false;
            } else if (ch == '"') c.quoted = true;
            else if (ch == '{') c.brackets++;
            else if (ch == '}') {
                c.brackets--;
                if (c.brackets == 0) {
                    std::string text;
                    text.swap(c.sb);
                    c.parsingJson = false;
                    dispatch(c, Json::parse_object(text));
                }
            }
            return;
        }
        if (ch == '\r') return;
        if (ch != '\n') {
            appendLatin1(c.sb, ch);
            return;
        }
        std::string parameters;
        parameters.swap(c.sb);
        if (parameters.find_first_not_of(" \t\n\v\f\r") == std::string::npos) return; // trim().length() == 0
        const std::string first = parameters.substr(0, parameters.find(';'));
        const int64_t dataSize = parseJavaInt(first);
        if (dataSize > 20 * 1024 * 1024) throw std::runtime_error("too large payload: " + std::to_string(dataSize));
        if (dataSize == 0) return;
        if (dataSize < 0) { // no size: assume JSON and read until its end
            c.parsingJson = true;
            c.stuffed = c.quoted = false;
            c.brackets = 0;
            return;
        }
        c.rawLeft = dataSize;
        c.raw.clear();
        c.raw.reserve(size_t(dataSize));
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
        return n;
    }
    void emulateToTime(Connection &c, int64_t time, int64_t timeId) // JSONClientConnection.java:326-353
    {
        std::vector<Node *> mine;
        for (Node *n : sim_.getNodes())
            if (connectionOf(n) == &c) mine.push_back(n);
        const std::vector<Info> info = nodeInfo(mine);
        if (!c.open()) return; // send() on a closed connection: nothing goes out
        // {"command":"time-step","id":..,"parameters":{"time":..,"node-info":[{..},..]}} -- written directly: this is
        // the one message whose size grows with the node count
        std::string &o = c.out;
        o += "{\"command\":\"time-step\",\"id\":";
        append_int(o, timeId);
        o += ",\"parameters\":{\"time\":";
        append_int(o, time);
        o += ",\"node-info\":[";
        for (size_t i = 0; i < mine.size(); ++i) {
            if (i) o += ',';
            o += "{\"node-id\":";
            Json::quote(mine[i]->getId(), o);
            o += ",\"rssi\":";
            append_double(o, info[i].rssi);
            o += ",\"receiving\":";
            append_int(o, info[i].receiving);
            o += ",\"wireless-channel\":";
            append_int(o, info[i].channel);
            o += '}';
        }
        o += "]}}";
        c.sent();
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
        for (Connection *e : em) emulateToTime(*e, time, waitingForTimeId_);
        usStepMessages_ += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
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
        for (const emul8::MediumCall &call : sim_.calls)
            if (call.kind == emul8::MediumCall::DELIVER) deliverRadioPacket(*call.packet, *call.destination, call.rssi);
        sim_.calls.clear();
        usMedium_ += std::chrono::duration<double, std::micro>(t1 - t0).count();
        usReceiveMessages_ += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
        ++steps_;
        prunePackets();
        if (timeController_) timeController_->send(Json::object().add("reply", Json::of("OK")).add("id", Json::of(timeControllerLastTimeId_)));
    }
    void deliverRadioPacket(const RadioPacket &p, Node &dst, double rssi) // :356-364 + RadioPacket.toJsonDestination
    {
        Connection *cc = connectionOf(&dst);
        if (!cc || !cc->connected) {
            VLOG("Node %s has no client connection", dst.getId().c_str());
            return;
        }
        // RadioPacket.toJsonDestination, written directly (tens of thousands per tick)
        std::string &o = cc->out;
        o += "{\"command\":\"receive\",\"node-id\":";
        Json::quote(dst.getId(), o);
        o += ",\"time-start\":";
        append_int(o, p.getStartTime());
        o += ",\"time-end\":";
        append_int(o, p.getEndTime());
        o += ",\"rf-power\":";
        append_double(o, rssi);
        o += ",\"wireless-channel\":";
        append_int(o, p.getWirelessChannel());
        o += ",\"packet-data\":";
        Json::quote(p.getPacketDataAsHex(), o);
        o += '}';
        cc->sent();
        ++deliveries_;
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

    void handleMessage(Connection &client, const Json &json)
    {
        int64_t time = sim_.getTime();
        if (json.get("reply")) { // getString("reply", null): a non-string member throws
            const std::string status = json.at("reply").asString();
            const int64_t id = json.getLong("id", -1);
            if (status == "OK") {
                if (id >= 0 && id == waitingForTimeId_) emulatorTimeStepped(client, id);
            } else {
                VLOG("%s error reply: %s", client.name.c_str(), json.toString().c_str());
            }
            return;
        }
        const int64_t id = json.getLong("id", -1);
        Json reply;
        bool haveReply = false, noreply = false;
        const Json *cmd = json.get("command");
        const std::string command = cmd ? cmd->asString() : std::string();
        if (!cmd) {
            reply = replyError(id, "command-error", "no command specified");
            haveReply = true;
        } else if (command == "time-get") {
            if (id >= 0) {
                reply = replyObject(id).set("reply-object", Json::object().add("time", Json::of(time)));
                haveReply = true;
            }
        } else if (command == "time-set") {
            if (id < 0) {
                reply = replyError(id, "command-error", "time-set must include reply id");
                haveReply = true;
            } else {
                if (!timeController_) timeController_ = &client;
                if (timeController_ == &client) {
                    try {
                        time = member(member(json, "parameters").asObject(), "time").asLong();
                        stepTime(time, id);
                        noreply = true;
                    } catch (const MissingMember &) { // a NullPointerException there: getMessage() is null
                        reply = replyError(id, "command-error", "failed to set time:null");
                        haveReply = true;
                    } catch (const std::exception &e) {
                        reply = replyError(id, "command-error", std::string("failed to set time:") + e.what());
                        haveReply = true;
                    }
                } else {
                    reply = replyError(id, "command-error", "only one time controller allowed");
                    haveReply = true;
                }
            }
        } else if (command == "transmit") {
            const std::string nodeId = member(json, "node-id").toString();
            const int64_t tTime = member(json, "time").asLong();
            const std::string packetData = json.getString("packet-data", "");
            Node *node = sim_.getNode(nodeId);
            if (!node) {
                VLOG("non-existing node sending radio packet: %s", nodeId.c_str());
                reply = replyError(id, "command-error", "could not find source node");
                haveReply = true;
            } else if (!medium_) {
                reply = replyError(id, "command-error", "no radio medium available");
                haveReply = true;
            } else {
                RadioPacket *owned = new RadioPacket(node, tTime, packetData);
                packets_.emplace(owned, std::unique_ptr<RadioPacket>(owned));
                RadioPacket &packet = *owned;
                const Json *value = json.get("rf-power");
                if (isNumber(value)) packet.setTransmitPower(value->asDouble());
                value = json.get("wireless-channel");
                if (isNumber(value)) packet.setWirelessChannel(value->asInt());
                sim_.notifyRadioListeners(packet);
                medium_->transmit(packet);
                mediumError("transmit");
                ++transmissions_;
            }
        } else if (command == "log") {
            const Json &params = member(json, "parameters").asObject();
            const std::string nodeId = member(params, "node-id").toString();
            const std::string logMsg = member(params, "message").asString();
            Node *node = sim_.getNode(nodeId);
            if (!node) throw std::runtime_error("log from a node that does not exist: " + nodeId); // node.log on null
            deliverLogEvent(*node, logMsg);
        } else if (command == "node-config-set") {
            const Json &params = member(json, "parameters").asObject();
            const std::string nodeId = member(params, "node-id").toString();
            settle(); // what was sent before this message saw the nodes as they were
            Node *node = addNode(nodeId, &client);
            const Json *value = params.get("position");
            if (value && value->isArray()) {
                const Json &p = *value;
                // (deviation: Double.parseDouble("1e999") is Infinity and the reference would keep it, leaving the node
                // unheard; the device-resident table takes finite coordinates only, so such a position is not applied)
                const double px = p.size() > 1 ? p[0].asDouble() : 0.0, py = p.size() > 1 ? p[1].asDouble() : 0.0;
                const double pz = p.size() > 2 ? p[2].asDouble() : 0.0;
                if (!std::isfinite(px) || !std::isfinite(py) || !std::isfinite(pz))
                    std::fprintf(stderr, "node %s: non-finite position ignored\n", nodeId.c_str());
                else if (p.size() > 2) node->getPosition().set(px, py, pz);
                else if (p.size() > 1) node->getPosition().set(px, py);
            }
            sim_.nodeChanged(node);
            value = params.get("rf-power");
            if (isNumber(value)) node->getRadio().setTransmitPower(value->asDouble());
            value = params.get("wireless-channel");
            if (isNumber(value)) node->getRadio().setWirelessChannel(value->asInt());
            value = params.get("rx-loss");
            if (isNumber(value)) node->getRadio().setRxProbability(value->asDouble());
            value = params.get("tx-loss");
            if (isNumber(value)) node->getRadio().setTxProbability(value->asDouble());
            value = params.get("radio-state");
            if (value && value->isString()) node->getRadio().setEnabled(value->asString() != "disabled");
            if (id >= 0) {
                const std::vector<Info> info = nodeInfo({node});
                Json nodeInfo = Json::object();
                nodeInfo.add("node-id", Json::of(nodeId));
                nodeInfo.add("rssi", Json::of(info[0].rssi));
                nodeInfo.add("receiving", Json::of(info[0].receiving));
                nodeInfo.add("wireless-channel", Json::of(info[0].channel));
                reply = replyObject(id).set("reply-object", Json::object().add("node-info", nodeInfo));
                haveReply = true;
            }
        } else if (command == "link-quality") {
            const Json &link = member(json, "link").asObject();
            (void)member(link, "src").toString();
            (void)member(link, "dst").toString();
            const Json *value = json.get("wireless-channel");
            if (isNumber(value)) (void)value->asInt();
            value = link.get("quality");
            if (isNumber(value)) (void)value->asInt(); // "TODO update radio medium" in the reference: parsed, not used
        } else if (command == "configuration-set") {
            if (timeController_) {
                reply = replyError(id, "command-error", "already initialized");
                haveReply = true;
            } else {
                const Json &params = member(json, "parameters").asObject();
                const Json *value = params.get("propagation-option");
                if (value) {
                    const std::string option = value->asString();
                    if (option == "n2n-link") {
                        const Json *matrix = params.get("matrix-data");
                        const Json *numberOfNodes = params.get("number-of-nodes");
                        if (!matrix || !matrix->isArray() || matrix->size() == 0) {
                            reply = replyError(id, "command-error", "no matrix specified");
                            haveReply = true;
                        } else if (member(params, "number-of-nodes").asInt() != int(std::sqrt(double(matrix->size())))) {
                            reply = replyError(id, "command-error", "inconsistent data matrix or nodes");
                            haveReply = true;
                        } else {
                            const int n = numberOfNodes->asInt();
                            std::vector<std::vector<double>> m;
                            m.assign(size_t(n), std::vector<double>(size_t(n), 0.0));
                            for (int i = 0; i < n; ++i)
                                for (int j = 0; j < n; ++j) m[size_t(i)][size_t(j)] = (*matrix)[size_t(j + i * n)].asDouble();
                            if (!opt_.noMedium) setMedium(new emul8::N2NRadioMedium(m, opt_.device));
                        }
                    } else if (option == "udgm") {
                        if (!opt_.noMedium) setMedium(new emul8::UDGMRadioMedium(opt_.device));
                    } else if (option == "udgm-constant-loss") { // (not wired to the protocol in the reference: its class exists)
                        if (!opt_.noMedium) setMedium(new emul8::UDGMConstantLossRadioMedium(opt_.device));
                    } else if (option == "log-distance") {
                        // the engine's extension medium: the only additions to the wire are this
                        // option string and its optional numeric parameters
                        rm_model_params p;
                        rm_model_defaults(&p, RM_MODEL_LOGDIST);
                        auto num = [&](const char *name, double &field) {
                            const Json *v = params.get(name);
                            if (isNumber(v)) field = v->asDouble();
                        };
                        num("reference-loss-db", p.ld_pl0_db);
                        num("path-loss-exponent", p.ld_exponent);
                        num("reference-distance", p.ld_d0);
                        num("shadowing-sigma-db", p.ld_sigma_db);
                        num("shadowing-clip", p.ld_clip);
                        num("sensitivity-dbm", p.ld_sensitivity_dbm);
                        num("noise-dbm", p.ld_noise_dbm);
                        num("capture-db", p.ld_capture_db);
                        num("interference-floor-dbm", p.ld_ifloor_dbm);
                        if (const Json *v = params.get("shadowing-seed"); isNumber(v)) p.ld_seed = uint64_t(v->asLong());
                        if (const Json *v = params.get("sinr"); v && v->type() == Json::BOOL && v->toString() == "true") p.flags |= RM_LD_SINR;
                        if (!opt_.noMedium) {
                            std::unique_ptr<emul8::LogDistanceRadioMedium> m(new emul8::LogDistanceRadioMedium(opt_.device));
                            m->params() = p;
                            try {
                                m->apply();
                                setMedium(m.release());
                            } catch (const std::exception &e) {
                                reply = replyError(id, "command-error", std::string("log-distance: ") + e.what());
                                haveReply = true;
                            }
                        }
                    } else if (option == "nullrm") {
                        // the null radio medium is the default
                    } else {
                        std::fprintf(stderr, "Unsupported propagation-option: %s - reverting to null radio medium\n", option.c_str());
                    }
                }
            }
        } else if (command == "subscribe-event") {
            eventListeners_.push_back(&client); // ArrayUtils.add: appended, duplicates allowed
        } else if (command == "unsubscribe-event") {
            auto it = std::find(eventListeners_.begin(), eventListeners_.end(), &client); // ArrayUtils.remove: the first one
            if (it != eventListeners_.end()) eventListeners_.erase(it);
        } else {
            reply = replyError(id, "command-error", "unsupported command: " + command);
            haveReply = true;
        }
        if (!haveReply && id >= 0 && !noreply) {
            reply = replyObject(id);
            haveReply = true;
        }
        if (haveReply) client.send(reply);
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
        std::fprintf(stderr,
                     "rsim_server: %llu steps, %llu transmissions, %llu deliveries, time %lld; per step: time-step messages %.1f us, "
                     "medium (tick + drain, deliveries on the host) %.1f us, receive messages %.1f us\n",
                     (unsigned long long)steps_, (unsigned long long)transmissions_, (unsigned long long)deliveries_,
                     (long long)sim_.getTime(), usStepMessages_ * k, usMedium_ * k, usReceiveMessages_ * k);
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
