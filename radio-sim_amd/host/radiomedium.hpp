// radiomedium.hpp -- C++ host-side mirror of the reference's radio-medium plug-in API, above the
// C ABI (include/radiomedium_hip.h).  Same class and method names, argument meaning and error
// behaviour as the reference's Java (paths relative to
// /root/reference/radio-medium/java/se/sics/emul8/radiomedium/), written from scratch:
//
//   Position            Position.java:35-65        x, y, z, set(x,y[,z]), getDistance
//   Transciever         Transciever.java:3-114     txpower 0.0, channel 26, enabled, rx/txProbability 1.0,
//                                                  getRSSI / getReceivingState / setReceiving / clearReceiving ...
//   Node                Node.java:39-99            id, getIdAsInteger (-1 if not numeric), position, radio
//   RadioPacket         RadioPacket.java:38-111    source, start time, txpower / channel copied from the source,
//                                                  getPacketAirTime = 32 us per hex character
//   Simulator           Simulator.java             only what a medium touches: addNode / getNodes / getNode,
//                                                  getTime, getRandom seed, generateTransmissionEvents,
//                                                  generateReceptionEvents, deliverRadioPacket (recorded, in call order)
//   RadioMedium         RadioMedium.java:35-45     getName / setSimulator / transmit / getBaseRSSI
//   AbstractRadioMedium AbstractRadioMedium.java   baseRSSI = -100.0, setBaseRSSI
//   NullRadioMedium, UDGMRadioMedium, UDGMConstantLossRadioMedium, N2NRadioMedium
//                       the four reference media, here backed by the MI355X engine: transmit() is one
//                       rm_transmit call, and the heard links -- returned in node order -- become exactly the
//                       Simulator calls the reference's loops make.
//
// Two more modes of the media, for hosts that batch (the reference consumes a tick's events only in
// emulatorTimeStepDone -> processAllEvents, Simulator.java:155-165, so nothing forces one evaluation per packet):
//   setTickMode(true)      transmit() only queues the packet; Simulator::emulatorTimeStepDone flushes the queue in
//                          ONE evaluation (rm_tick_begin / rm_enqueue_tx / rm_tick_flush_view) before the time moves,
//                          and the medium then makes the very calls of the per-packet mode, in arrival x node order.
//   setDeviceEvents(true)  the events never reach the host: the engine keeps packets, heard links and every
//                          node's radio state on the device (rm_events_*), Simulator::emulatorTimeStepDone hands
//                          out the drain's deliveries in the reference queue's order, nodeInfo() the time-step
//                          message's per-node fields.
//
// There is no CPU evaluation in this file: without a gfx950 device the medium's constructor throws.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <deque>
#include <memory>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/radiomedium_hip.h"

namespace emul8 {

class Simulator;
class RadioMedium;
class RadioPacket;

class Position {
public:
    double x = 0, y = 0, z = 0;
    void set(double x_, double y_) { set(x_, y_, 0.0); }
    void set(double x_, double y_, double z_) { x = x_; y = y_; z = z_; }
};

class Node;

class Transciever {
public:
    static constexpr int LISTENING = 0, TRANSMITTING = 1, RECEIVING = 2, DISABLED = 3;
    explicit Transciever(Node *node) : node_(node) {}
    Node *getNode() const { return node_; }
    bool isEnabled() const { return enabled_; }
    void setEnabled(bool e) { enabled_ = e; }
    double getTransmitPower() const { return txpower_; }
    void setTransmitPower(double p) { txpower_ = p; }
    int getWirelessChannel() const { return channel_; }
    void setWirelessChannel(int c) { channel_ = c; }
    double getRxProbability() const { return rxProbability_; }
    void setRxProbability(double p) { rxProbability_ = p; }
    double getTxProbability() const { return txProbability_; }
    void setTxProbability(double p) { txProbability_ = p; }
    bool isReceiving() const { return receiving_ != nullptr; }
    double getRSSI() const; // latched rssi while receiving, else the medium's base RSSI, else -99.99
    int getReceivingState() const
    {
        if (!enabled_) return DISABLED;
        if (receiving_) return RECEIVING;
        if (sending_) return TRANSMITTING;
        return LISTENING;
    }
    void setReceiving(const RadioPacket *p, double rssi) { clearSending(); receiving_ = p; receivingRSSI_ = rssi; }
    void clearReceiving() { receiving_ = nullptr; }
    void setSending(const RadioPacket *p) { clearReceiving(); sending_ = p; }
    void clearSending() { sending_ = nullptr; }

private:
    Node *node_;
    double txpower_ = 0.0;
    int channel_ = 26;
    bool enabled_ = true;
    const RadioPacket *receiving_ = nullptr, *sending_ = nullptr;
    double receivingRSSI_ = 0.0;
    double rxProbability_ = 1.0, txProbability_ = 1.0;
};

class Node {
public:
    Node(const std::string &id, Simulator *sim) : id_(id), sim_(sim), radio_(this)
    {
        char *end = nullptr;
        const long v = std::strtol(id.c_str(), &end, 10);
        intID_ = (!id.empty() && end && *end == '\0') ? int(v) : -1; // Integer.parseInt failed -> -1
    }
    const std::string &getId() const { return id_; }
    int getIdAsInteger() const { return intID_; }
    Position &getPosition() { return pos_; }
    Transciever &getRadio() { return radio_; }
    Simulator *getSimulator() const { return sim_; }
    RadioMedium *getRadioMedium() const;
    int index = -1; // position in Simulator.getNodes() (registration order)

private:
    std::string id_;
    int intID_;
    Simulator *sim_;
    Position pos_;
    Transciever radio_;
};

class RadioPacket {
public:
    RadioPacket(Node *node, int64_t time, const std::string &packetData)
        : node_(node), time_(time), txpower_(node->getRadio().getTransmitPower()),
          channel_(node->getRadio().getWirelessChannel()), data_(packetData) {}
    Node *getSource() const { return node_; }
    int64_t getStartTime() const { return time_; }
    int64_t getEndTime() const { return time_ + getPacketAirTime(); }
    int64_t getPacketAirTime() const { return int64_t(data_.size()) * 32; }
    double getTransmitPower() const { return txpower_; }
    void setTransmitPower(double p) { txpower_ = p; }
    int getWirelessChannel() const { return channel_; }
    void setWirelessChannel(int c) { channel_ = c; }
    const std::string &getPacketDataAsHex() const { return data_; }
    // RadioPacket.java:97-99 (DatatypeConverter.parseHexBinary): two hex digits per byte
    std::vector<uint8_t> getPacketDataAsBytes() const
    {
        if (data_.size() % 2) throw std::invalid_argument("hexBinary needs to be even-length: " + data_);
        auto nib = [&](char ch) -> int {
            if (ch >= '0' && ch <= '9') return ch - '0';
            if (ch >= 'a' && ch <= 'f') return ch - 'a' + 10;
            if (ch >= 'A' && ch <= 'F') return ch - 'A' + 10;
            throw std::invalid_argument("contains illegal character for hexBinary: " + data_);
        };
        std::vector<uint8_t> out(data_.size() / 2);
        for (size_t i = 0; i < out.size(); ++i) out[i] = uint8_t(nib(data_[2 * i]) * 16 + nib(data_[2 * i + 1]));
        return out;
    }

private:
    Node *node_;
    int64_t time_;
    double txpower_;
    int channel_;
    std::string data_;
};

class RadioMedium {
public:
    virtual ~RadioMedium() = default;
    virtual std::string getName() = 0;
    virtual void setSimulator(Simulator *sim) = 0;
    virtual void transmit(RadioPacket &packet) = 0;
    virtual double getBaseRSSI(Node &node) = 0;
    // hooks of Simulator::emulatorTimeStepDone (not in the reference's interface; no-ops for a plain medium)
    virtual void flush() {}                       // evaluate the transmit() calls queued since the last tick end
    virtual void processEvents(int64_t /*time*/) {} // the tick-end drain, when the medium keeps the events itself
};

// what the medium hands back to the simulation core, recorded in call order
struct MediumCall {
    enum Kind { TRANSMISSION_EVENTS, RECEPTION_EVENTS, DELIVER } kind;
    const RadioPacket *packet;
    Node *destination;   // null for TRANSMISSION_EVENTS
    double rssi;
    bool doDeliver;
    int64_t timeStart, timeEnd; // Simulator.java:323-333: max(start, currentTime), + air time
};

// RadioListener.java:35-38: "will receive all packets transmitted during a simulation"
class RadioListener {
public:
    virtual ~RadioListener() {}
    virtual void packetTransmission(RadioPacket &packet) = 0;
};

class Simulator {
public:
    explicit Simulator(int64_t randomSeed = 0) : seed_(randomSeed) {}
    // Simulator.java:200-211; the JSON handler notifies the listeners right after medium.transmit
    // (net/SimulatorJSONHandler.java:92)
    void addRadioListener(RadioListener *l) { listeners_.push_back(l); }
    void notifyRadioListeners(RadioPacket &p)
    {
        for (RadioListener *l : listeners_) l->packetTransmission(p);
    }
    int64_t getRandomSeed() const { return seed_; }
    int64_t getTime() const { return currentTime_; }
    void setTime(int64_t t) { currentTime_ = t; }
    RadioMedium *getRadioMedium() const { return medium_; }
    void setRadioMedium(RadioMedium *m)
    {
        if (m) m->setSimulator(this);
        medium_ = m;
    }
    Node *getNode(const std::string &id)
    {
        auto it = table_.find(id);
        return it == table_.end() ? nullptr : it->second;
    }
    const std::vector<Node *> &getNodes() const { return nodes_; }
    uint64_t nodesVersion() const { return version_; }
    void nodesChanged() { ++version_; } // call after changing fields of existing nodes (no hook in the reference)
    // the finer hook: one node's fields changed (node-config-set, SimulatorJSONHandler.java:105-143);
    // the medium writes just these nodes to the device before its next evaluation (rm_node_update)
    void nodeChanged(const Node *n) { if (n) dirty_.push_back(n->index); }
    std::vector<int> takeChangedNodes()
    {
        std::vector<int> d;
        d.swap(dirty_);
        return d;
    }
    Node *addNode(const std::string &id)
    {
        if (Node *n = getNode(id)) return n; // already handled
        owned_.emplace_back(new Node(id, this));
        Node *n = owned_.back().get();
        n->index = int(nodes_.size());
        nodes_.push_back(n);
        table_[id] = n;
        ++version_;
        return n;
    }
    // Simulator.emulatorTimeStepDone (Simulator.java:155-165): currentTime = stepTime; processAllEvents(currentTime).
    // A medium in tick mode evaluates its queued transmit() calls first -- while currentTime is still the old one,
    // which is what the event times of those packets were computed with in the reference (:323-326).
    void emulatorTimeStepDone(int64_t stepTime)
    {
        if (medium_) medium_->flush();
        currentTime_ = stepTime;
        if (medium_) medium_->processEvents(stepTime);
    }
    void generateTransmissionEvents(RadioPacket &p) { record(MediumCall::TRANSMISSION_EVENTS, p, nullptr, 0.0, false); }
    void generateReceptionEvents(RadioPacket &p, Node *dst, double rssi, bool doDeliver)
    {
        record(MediumCall::RECEPTION_EVENTS, p, dst, rssi, doDeliver);
    }
    void deliverRadioPacket(RadioPacket &p, Node *dst, double rssi) { record(MediumCall::DELIVER, p, dst, rssi, true); }
    std::vector<MediumCall> calls;
    bool recording = true;   // false: the calls are only counted (timing runs: the list is this stub's, not the engine's, work)
    uint64_t callCount = 0;

private:
    void record(MediumCall::Kind k, RadioPacket &p, Node *dst, double rssi, bool deliver)
    {
        ++callCount;
        if (!recording) return;
        int64_t t0 = p.getStartTime();
        if (t0 < currentTime_) t0 = currentTime_;
        calls.push_back({k, &p, dst, rssi, deliver, t0, t0 + p.getPacketAirTime()});
    }
    std::vector<RadioListener *> listeners_;
    int64_t seed_;
    int64_t currentTime_ = 0;
    RadioMedium *medium_ = nullptr;
    std::vector<std::unique_ptr<Node>> owned_;
    std::vector<Node *> nodes_;
    std::unordered_map<std::string, Node *> table_;
    uint64_t version_ = 0;
    std::vector<int> dirty_;
};

// The receiver-side state machine the verdicts drive (events/ReceptionEvent.java:12-46,
// events/TransmissionEvent.java:18-26): the start flank latches packet + rssi (and clears a pending
// transmission), the end flank clears the reception -- whichever packet it belongs to -- and
// delivers only in delivery mode.
enum class ReceptionMode { start, interference, delivery };

struct ReceptionEvent {
    int64_t time;
    Simulator *simulator;
    RadioPacket *packet;
    Node *destination;
    double rssi;
    ReceptionMode mode;
    void execute(int64_t /*currentTime*/)
    {
        Transciever &t = destination->getRadio();
        if (mode == ReceptionMode::start) {
            t.setReceiving(packet, rssi);
        } else {
            t.clearReceiving();
            if (mode == ReceptionMode::delivery) simulator->deliverRadioPacket(*packet, destination, rssi);
        }
    }
};

struct TransmissionEvent {
    int64_t time;
    RadioPacket *packet;
    bool isStart;
    void execute(int64_t /*currentTime*/)
    {
        Transciever &t = packet->getSource()->getRadio();
        if (isStart) t.setSending(packet);
        else t.clearSending();
    }
};

// the two events generateReceptionEvents queues for one heard link (Simulator.java:321-335)
inline void makeReceptionEvents(Simulator &sim, const MediumCall &c, RadioPacket &p, ReceptionEvent &startEv, ReceptionEvent &endEv)
{
    startEv = {c.timeStart, &sim, &p, c.destination, c.rssi, ReceptionMode::start};
    endEv = {c.timeEnd, &sim, &p, c.destination, c.rssi, c.doDeliver ? ReceptionMode::delivery : ReceptionMode::interference};
}

inline RadioMedium *Node::getRadioMedium() const { return sim_->getRadioMedium(); }
inline double Transciever::getRSSI() const
{
    if (isReceiving()) return receivingRSSI_;
    RadioMedium *m = node_->getRadioMedium();
    return m ? m->getBaseRSSI(*node_) : -99.99;
}

class AbstractRadioMedium : public RadioMedium {
public:
    void setSimulator(Simulator *sim) override { simulator = sim; }
    double getBaseRSSI(Node &) override { return baseRSSI; }
    void setBaseRSSI(double rssi) { baseRSSI = rssi; }

protected:
    Simulator *simulator = nullptr;
    double baseRSSI = -100.0;
};

// One MI355X context behind the RadioMedium contract.
class GpuRadioMedium : public AbstractRadioMedium {
public:
    explicit GpuRadioMedium(int kind, int device = 0) : kind_(kind)
    {
        if (rm_abi_version() != RM_ABI_VERSION)
            throw std::runtime_error("libradiomedium_hip.so does not have the ABI version of the header this host was built against");
        if (rm_create(device, &ctx_) != RM_OK) throw std::runtime_error(std::string("no MI355X radio medium: ") + rm_last_error());
        rm_model_defaults(&params_, kind);
        apply();
    }
    ~GpuRadioMedium() override { rm_destroy(ctx_); }
    GpuRadioMedium(const GpuRadioMedium &) = delete;
    GpuRadioMedium &operator=(const GpuRadioMedium &) = delete;

    std::string getName() override { return rm_get_name(ctx_); }
    void setSimulator(Simulator *sim) override
    {
        simulator = sim;
        if (sim) rm_seed(ctx_, sim->getRandomSeed()); // Simulator.getRandom(): one generator for all packets
        uploaded_ = ~0ull;
    }
    double getBaseRSSI(Node &n) override { return rm_get_base_rssi(ctx_, n.index); }
    void setBaseRSSI(double rssi)
    {
        baseRSSI = rssi;
        rm_set_base_rssi(ctx_, rssi);
    }

    // ---- batching modes (see the top of this file)
    void setTickMode(bool on) { tickMode_ = on; }
    bool getTickMode() const { return tickMode_; }
    void setDeviceEvents(bool on)
    {
        if (on == deviceEvents_) return;
        if ((on ? rm_events_enable(ctx_, 0, 0) : rm_events_disable(ctx_)) != RM_OK) throw std::runtime_error(rm_last_error());
        deviceEvents_ = on;
        for (size_t i = inFlightHead_; i < inFlight_.size(); ++i) released_.push_back(inFlight_[i]);
        inFlight_.clear();
        inFlightHead_ = 0;
        firstInFlight_ = on ? rm_events_next_packet(ctx_) : 0;
    }
    bool getDeviceEvents() const { return deviceEvents_; }
    // Tick mode / device events: the medium refers to a RadioPacket from transmit() until it says so here -- the
    // packet's last event has fired on the device, or a failed evaluation dropped it.  A host that owns the
    // RadioPacket objects frees exactly these (by identity: a failed flush leaves no packet behind that an older,
    // still pending one could be mistaken for).  Without device events the packets handed to transmit() live in the
    // host Simulator's own events, as in the reference.
    void takeReleased(std::vector<RadioPacket *> &out)
    {
        out.insert(out.end(), released_.begin(), released_.end());
        released_.clear();
    }
    size_t inFlightCount() const { return queue_.size() + inFlight_.size() - inFlightHead_; }

    // the transmit() calls queued in tick mode, in ONE evaluation; then the calls the per-packet mode makes
    void flush() override
    {
        if (queue_.empty()) return;
        lastError.clear();
        Simulator *sim = simulator;
        std::vector<RadioPacket *> q;
        q.swap(queue_);
        // whatever fails from here on: the tick's packets were never numbered by the engine, nothing refers to them
        auto dropped = [&](const std::string &why) {
            lastError = why;
            if (deviceEvents_) released_.insert(released_.end(), q.begin(), q.end());
        };
        if (!sim) return dropped("No simulator");
        const std::vector<Node *> &nodes = sim->getNodes();
        if (!sync(sim, nodes)) return dropped(lastError);
        rm_set_time(ctx_, sim->getTime());
        int rc = rm_tick_begin(ctx_, sim->getTime(), sim->getTime());
        for (size_t k = 0; k < q.size() && rc == RM_OK; ++k) {
            const double txp = q[k]->getTransmitPower();
            const int32_t ch = q[k]->getWirelessChannel();
            rc = rm_enqueue_tx(ctx_, q[k]->getSource()->index, q[k]->getStartTime(), q[k]->getPacketAirTime(), &txp, &ch);
        }
        if (rc != RM_OK) return dropped(rm_last_error());
        if (deviceEvents_) { // nothing comes back: packets, links and events stay on the device
            const int64_t before = rm_events_next_packet(ctx_);
            if (rm_tick_run(ctx_) != RM_OK) return dropped(rm_last_error());
            // the engine numbers the packets of a tick it has taken, in enqueue order
            if (rm_events_next_packet(ctx_) - before != int64_t(q.size()) || before != firstInFlight_ + int64_t(inFlight_.size() - inFlightHead_))
                return dropped("radio medium: packet numbers out of step with the engine");
            for (RadioPacket *p : q) inFlight_.push_back(p);
            return;
        }
        rm_host_result r{};
        if (rm_tick_flush_view(ctx_, &r) != RM_OK) { lastError = rm_last_error(); return; }
        for (uint32_t k = 0; k < r.n_packets; ++k) { // arrival order, then node order: the per-packet calls
            RadioPacket &packet = *q[k];
            if (kind_ != RM_MODEL_UDGM_CONST) sim->generateTransmissionEvents(packet);
            for (uint32_t i = r.pkt_offset[k]; i < r.pkt_offset[k + 1]; ++i) {
                Node *node = nodes[size_t(r.dst[i])];
                // (the reference's media hand the packet's transmit power through -- UDGMRadioMedium.java:95 --: one value per packet)
                const double rssi = r.rssi ? r.rssi[i] : r.pkt_rssi[k];
                if (kind_ == RM_MODEL_UDGM_CONST) sim->deliverRadioPacket(packet, node, rssi);
                else sim->generateReceptionEvents(packet, node, rssi, r.verdict[i] == RM_DELIVERED);
            }
        }
    }

    // device events: Simulator.processAllEvents(time) as one drain on the device; the deliveries come back in the
    // order the reference's queue pops them and become Simulator.deliverRadioPacket calls (ReceptionEvent.java:41-44)
    void processEvents(int64_t time) override
    {
        if (!deviceEvents_) return;
        Simulator *sim = simulator;
        rm_delivery_view v{};
        if (rm_events_process(ctx_, time, &v) != RM_OK) { lastError = rm_last_error(); return; }
        const std::vector<Node *> &nodes = sim->getNodes();
        RadioPacket *const *const flying = inFlight_.data() + inFlightHead_;
        const size_t n_flying = inFlight_.size() - inFlightHead_;
        for (uint32_t r = 0; r < v.n_runs; ++r) { // a run = the deliveries of one packet (adjacent in the queue's pop order)
            const int64_t k = v.run_packet[r] - firstInFlight_;
            const uint32_t first = v.run_first[r], end = first + v.run_count[r];
            if (k < 0 || size_t(k) >= n_flying || end > v.count || end < first) {
                lastError = "radio medium: a delivery names a packet the host does not hold";
                continue;
            }
            for (uint32_t i = first; i < end; ++i) {
                if (v.dst[i] < 0 || size_t(v.dst[i]) >= nodes.size()) {
                    lastError = "radio medium: a delivery names a node the host does not hold";
                    continue;
                }
                sim->deliverRadioPacket(*flying[size_t(k)], nodes[size_t(v.dst[i])], v.rssi[i]);
            }
        }
        // packets whose last event has fired are forgotten: everything below the oldest number still queued
        while (firstInFlight_ < v.oldest_packet && inFlightHead_ < inFlight_.size()) {
            released_.push_back(inFlight_[inFlightHead_]);
            ++inFlightHead_;
            ++firstInFlight_;
        }
        if (inFlightHead_ > 4096 && inFlightHead_ * 2 > inFlight_.size()) { // the consumed front half goes away
            inFlight_.erase(inFlight_.begin(), inFlight_.begin() + long(inFlightHead_));
            inFlightHead_ = 0;
        }
    }

    // the per-node fields of a time-step message (net/JSONClientConnection.java:331-341) from the device's radio state
    bool nodeInfo(const std::vector<int32_t> &nodes, std::vector<double> &rssi, std::vector<int32_t> &receiving,
                  std::vector<int32_t> &channel)
    {
        rssi.resize(nodes.size()); receiving.resize(nodes.size()); channel.resize(nodes.size());
        if (simulator && !sync(simulator, simulator->getNodes())) return false; // the device mirrors the node table first
        if (rm_node_info(ctx_, nodes.data(), int32_t(nodes.size()), rssi.data(), receiving.data(), channel.data()) != RM_OK) {
            lastError = rm_last_error();
            return false;
        }
        return true;
    }

    // ... and only of the nodes whose fields differ from what this call reported for them last (every node the first time,
    // and after the node table has changed size): a host that keeps the text of its last time-step message re-serialises these
    bool nodeInfoChanged(std::vector<int32_t> &nodes, std::vector<double> &rssi, std::vector<int32_t> &receiving, std::vector<int32_t> &channel)
    {
        nodes.clear(); rssi.clear(); receiving.clear(); channel.clear();
        if (!simulator) return true;
        const size_t n = simulator->getNodes().size();
        if (!sync(simulator, simulator->getNodes())) return false; // the device mirrors the node table first
        if (n == 0) return true;
        // room for every node, kept between the calls (a step's answer is a few of them: four fresh vectors of n elements --
        // allocated, zeroed, paged in -- cost more than the query itself at 100 k nodes)
        if (chgNodes_.size() < n) {
            chgNodes_.resize(n); chgRssi_.resize(n); chgRecv_.resize(n); chgChan_.resize(n);
        }
        int32_t count = 0;
        if (rm_node_info_changed(ctx_, chgNodes_.data(), chgRssi_.data(), chgRecv_.data(), chgChan_.data(), int32_t(n), &count) != RM_OK) {
            lastError = rm_last_error();
            return false;
        }
        nodes.assign(chgNodes_.begin(), chgNodes_.begin() + count);
        rssi.assign(chgRssi_.begin(), chgRssi_.begin() + count);
        receiving.assign(chgRecv_.begin(), chgRecv_.begin() + count);
        channel.assign(chgChan_.begin(), chgChan_.begin() + count);
        return true;
    }

    // transmit(): never throws (the reference's returns void); failures are reported through lastError
    void transmit(RadioPacket &packet) override
    {
        if (tickMode_) { // evaluated at the end of the tick, with everything else that was sent in it
            queue_.push_back(&packet);
            return;
        }
        lastError.clear();
        Simulator *sim = simulator;
        if (!sim) { lastError = "No simulator"; return; }
        const std::vector<Node *> &nodes = sim->getNodes();
        if (!sync(sim, nodes)) return;
        rm_set_time(ctx_, sim->getTime());
        const double txp = packet.getTransmitPower();
        const int32_t ch = packet.getWirelessChannel();
        uint32_t heard = 0;
        uint8_t interference = 0;
        const int64_t before = deviceEvents_ ? rm_events_next_packet(ctx_) : 0;
        const int rc = rm_transmit(ctx_, packet.getSource()->index, packet.getStartTime(),
                                   int64_t(packet.getPacketDataAsHex().size()), &txp, &ch, dst_.data(), verdict_.data(),
                                   rssi_.data(), sinr_.data(), uint32_t(dst_.size()), &heard, &interference);
        if (deviceEvents_) { // the engine queued the packet's events itself -- if it numbered the packet
            if (rm_events_next_packet(ctx_) == before + 1 && before == firstInFlight_ + int64_t(inFlight_.size() - inFlightHead_))
                inFlight_.push_back(&packet);
            else
                released_.push_back(&packet);
            if (rc != RM_OK) lastError = rm_last_error();
            else lastInterference = interference != 0;
            return;
        }
        if (rc != RM_OK) { lastError = rm_last_error(); return; }
        lastInterference = interference != 0;
        if (kind_ != RM_MODEL_UDGM_CONST) sim->generateTransmissionEvents(packet);
        for (uint32_t i = 0; i < heard; ++i) { // node order, as the reference's for (Node node : nodes)
            Node *node = nodes[dst_[i]];
            if (kind_ == RM_MODEL_UDGM_CONST) sim->deliverRadioPacket(packet, node, rssi_[i]);
            else sim->generateReceptionEvents(packet, node, rssi_[i], verdict_[i] == RM_DELIVERED);
        }
    }
    std::string lastError;
    bool lastInterference = false;

protected:
    rm_model_params params_{};
    void apply()
    {
        if (rm_set_model(ctx_, &params_) != RM_OK) throw std::invalid_argument(rm_last_error());
    }
    rm_context *ctx_ = nullptr;

private:
    bool sync(Simulator *sim, const std::vector<Node *> &nodes)
    {
        if (uploaded_ == sim->nodesVersion()) {
            for (int i : sim->takeChangedNodes()) { // the dirty list: only these nodes go to the device
                Node &nd = *nodes[size_t(i)];
                if (rm_node_update(ctx_, i, nd.getPosition().x, nd.getPosition().y, nd.getPosition().z,
                                   nd.getRadio().getTransmitPower(), nd.getRadio().getWirelessChannel(),
                                   nd.getRadio().isEnabled(), nd.getRadio().getRxProbability(),
                                   nd.getRadio().getTxProbability()) != RM_OK) {
                    lastError = rm_last_error();
                    uploaded_ = ~0ull; // the rest of the dirty list is gone with this call: a fresh snapshot next time
                    return false;
                }
            }
            return true;
        }
        (void)sim->takeChangedNodes(); // covered by the snapshot
        const size_t n = nodes.size();
        std::vector<double> x(n), y(n), z(n), tp(n), rp(n), xp(n);
        std::vector<int32_t> ch(n), id(n);
        std::vector<uint8_t> en(n);
        for (size_t i = 0; i < n; ++i) {
            Node &nd = *nodes[i];
            x[i] = nd.getPosition().x; y[i] = nd.getPosition().y; z[i] = nd.getPosition().z;
            tp[i] = nd.getRadio().getTransmitPower(); ch[i] = nd.getRadio().getWirelessChannel();
            en[i] = nd.getRadio().isEnabled(); rp[i] = nd.getRadio().getRxProbability();
            xp[i] = nd.getRadio().getTxProbability(); id[i] = nd.getIdAsInteger();
        }
        if (rm_nodes_upload(ctx_, int32_t(n), x.data(), y.data(), z.data(), tp.data(), ch.data(), en.data(), rp.data(),
                            xp.data(), id.data()) != RM_OK) {
            lastError = rm_last_error();
            return false;
        }
        dst_.resize(n + 1); verdict_.resize(n + 1); rssi_.resize(n + 1); sinr_.resize(n + 1);
        uploaded_ = sim->nodesVersion();
        return true;
    }
    std::vector<int32_t> chgNodes_, chgRecv_, chgChan_; // nodeInfoChanged: room for every node
    std::vector<double> chgRssi_;
    int kind_;
    bool tickMode_ = false, deviceEvents_ = false;
    std::vector<RadioPacket *> queue_;   // tick mode: transmit() calls since the last flush
    std::vector<RadioPacket *> inFlight_; // device events: packets with events still queued, by packet number (from inFlightHead_)
    std::vector<RadioPacket *> released_; // packets the medium has stopped referring to since the last takeReleased()
    size_t inFlightHead_ = 0;
    int64_t firstInFlight_ = 0;
    uint64_t uploaded_ = ~0ull;
    std::vector<int32_t> dst_;
    std::vector<uint8_t> verdict_;
    std::vector<double> rssi_, sinr_;
};

class NullRadioMedium : public GpuRadioMedium {
public:
    explicit NullRadioMedium(int device = 0) : GpuRadioMedium(RM_MODEL_NULL, device) {}
};

class UDGMRadioMedium : public GpuRadioMedium {
public:
    explicit UDGMRadioMedium(int device = 0) : GpuRadioMedium(RM_MODEL_UDGM, device) {}
    double getSuccessRatioTx() const { return params_.udgm_success_ratio_tx; }
    void setSuccessRatioTx(double v) { params_.udgm_success_ratio_tx = v; apply(); }
    double getSuccessRatioRx() const { return params_.udgm_success_ratio_rx; }
    void setSuccessRatioRx(double v) { params_.udgm_success_ratio_rx = v; apply(); }
    double getTransmissionRange() const { return params_.udgm_transmission_range; }
    void setTransmissionRange(double v) { params_.udgm_transmission_range = v; apply(); }
    double getInterferenceRange() const { return params_.udgm_interference_range; }
    void setInterferenceRange(double v) { params_.udgm_interference_range = v; apply(); }
};

class UDGMConstantLossRadioMedium : public GpuRadioMedium {
public:
    explicit UDGMConstantLossRadioMedium(int device = 0) : GpuRadioMedium(RM_MODEL_UDGM_CONST, device) {}
};

class N2NRadioMedium : public GpuRadioMedium {
public:
    explicit N2NRadioMedium(const std::vector<std::vector<double>> &m, int device = 0) : GpuRadioMedium(RM_MODEL_N2N, device)
    {
        setMatrix(m);
    }
    void setMatrix(const std::vector<std::vector<double>> &m)
    {
        // The reference bounds the source id by the number of rows and the destination id by THAT row's length
        // (N2NRadioMedium.java:28-37), anything outside gives probability 0 = unheard: a jagged matrix is the square
        // matrix of side max(rows, longest row) padded with zeros.
        size_t n = m.size();
        for (const auto &row : m) n = std::max(n, row.size());
        std::vector<double> flat(n * n, 0.0);
        for (size_t i = 0; i < m.size(); ++i)
            for (size_t j = 0; j < m[i].size(); ++j) flat[i * n + j] = m[i][j];
        if (rm_set_n2n_matrix(ctx_, int32_t(n), flat.data()) != RM_OK) throw std::invalid_argument(rm_last_error());
    }
};

// The build's extension medium behind the same contract (NOT a reference class): log-distance path
// loss, per-link log-normal shadowing, and with setSinr(true) co-channel SINR capture + half duplex over the frames on
// the air.  With SINR the verdicts of a tick's frames depend on each other: tick mode decides them against everything
// sent in the tick, the per-packet mode against what was sent before -- the reference media do not have this coupling.
class LogDistanceRadioMedium : public GpuRadioMedium {
public:
    explicit LogDistanceRadioMedium(int device = 0) : GpuRadioMedium(RM_MODEL_LOGDIST, device) {}
    rm_model_params &params() { return params_; } // change, then apply()
    using GpuRadioMedium::apply;
    void setSinr(bool on)
    {
        params_.flags = on ? (params_.flags | RM_LD_SINR) : (params_.flags & ~RM_LD_SINR);
        apply();
    }
};

// The same contract over SEVERAL devices (or several partitions of one device): one rm_group behind one medium
// object, because the reference host is one process (Main.java:65-73).  The receivers are range-partitioned over the
// members; a packet is handed to every member, the members' heard links come back merged in node order, and the
// medium makes the calls of the single-device media.
class GroupRadioMedium : public AbstractRadioMedium {
public:
    GroupRadioMedium(int kind, const std::vector<int32_t> &devices) : kind_(kind)
    {
        if (rm_group_create(int32_t(devices.size()), devices.data(), &grp_) != RM_OK)
            throw std::runtime_error(std::string("no MI355X radio medium group: ") + rm_last_error());
        rm_model_defaults(&params_, kind);
        apply();
    }
    ~GroupRadioMedium() override { rm_group_destroy(grp_); }
    GroupRadioMedium(const GroupRadioMedium &) = delete;
    GroupRadioMedium &operator=(const GroupRadioMedium &) = delete;

    std::string getName() override { return rm_get_name(rm_group_context(grp_, 0)); }
    void setSimulator(Simulator *sim) override
    {
        simulator = sim;
        if (sim) rm_group_seed(grp_, sim->getRandomSeed());
        uploaded_ = ~0ull;
    }
    rm_model_params &params() { return params_; } // change, then apply()
    void apply()
    {
        if (rm_group_set_model(grp_, &params_) != RM_OK) throw std::invalid_argument(rm_last_error());
    }
    void setMatrix(const std::vector<std::vector<double>> &m) // N2NRadioMedium(double[][]): jagged rows as a zero-padded square
    {
        size_t side = m.size();
        for (const auto &row : m) side = std::max(side, row.size());
        std::vector<double> flat(side * side, 0.0);
        for (size_t i = 0; i < m.size(); ++i)
            for (size_t j = 0; j < m[i].size(); ++j) flat[i * side + j] = m[i][j];
        if (rm_group_set_n2n_matrix(grp_, int32_t(side), flat.data()) != RM_OK) throw std::invalid_argument(rm_last_error());
    }
    void transmit(RadioPacket &packet) override
    {
        lastError.clear();
        Simulator *sim = simulator;
        if (!sim) { lastError = "No simulator"; return; }
        const std::vector<Node *> &nodes = sim->getNodes();
        if (!sync(sim, nodes)) return;
        rm_group_set_time(grp_, sim->getTime());
        const double txp = packet.getTransmitPower();
        const int32_t ch = packet.getWirelessChannel();
        uint32_t heard = 0;
        uint8_t interference = 0;
        int rc = rm_group_tick_begin(grp_, packet.getStartTime(), packet.getStartTime());
        if (rc == RM_OK) rc = rm_group_enqueue_tx(grp_, packet.getSource()->index, packet.getStartTime(), packet.getPacketAirTime(), &txp, &ch);
        if (rc == RM_OK)
            rc = rm_group_tick_flush(grp_, nullptr, dst_.data(), verdict_.data(), rssi_.data(), nullptr, uint32_t(dst_.size()), &heard,
                                     &interference, nullptr);
        if (rc != RM_OK) { lastError = rm_last_error(); return; }
        if (kind_ != RM_MODEL_UDGM_CONST) sim->generateTransmissionEvents(packet);
        for (uint32_t i = 0; i < heard; ++i) {
            Node *node = nodes[size_t(dst_[i])];
            if (kind_ == RM_MODEL_UDGM_CONST) sim->deliverRadioPacket(packet, node, rssi_[i]);
            else sim->generateReceptionEvents(packet, node, rssi_[i], verdict_[i] == RM_DELIVERED);
        }
    }
    std::string lastError;

private:
    bool sync(Simulator *sim, const std::vector<Node *> &nodes)
    {
        if (uploaded_ == sim->nodesVersion()) {
            for (int i : sim->takeChangedNodes()) {
                Node &nd = *nodes[size_t(i)];
                if (rm_group_node_update(grp_, i, nd.getPosition().x, nd.getPosition().y, nd.getPosition().z, nd.getRadio().getTransmitPower(),
                                         nd.getRadio().getWirelessChannel(), nd.getRadio().isEnabled(), nd.getRadio().getRxProbability(),
                                         nd.getRadio().getTxProbability()) != RM_OK) {
                    lastError = rm_last_error();
                    return false;
                }
            }
            return true;
        }
        (void)sim->takeChangedNodes();
        const size_t n = nodes.size();
        std::vector<double> x(n), y(n), z(n), tp(n), rp(n), xp(n);
        std::vector<int32_t> ch(n), id(n);
        std::vector<uint8_t> en(n);
        for (size_t i = 0; i < n; ++i) {
            Node &nd = *nodes[i];
            x[i] = nd.getPosition().x; y[i] = nd.getPosition().y; z[i] = nd.getPosition().z;
            tp[i] = nd.getRadio().getTransmitPower(); ch[i] = nd.getRadio().getWirelessChannel();
            en[i] = nd.getRadio().isEnabled(); rp[i] = nd.getRadio().getRxProbability();
            xp[i] = nd.getRadio().getTxProbability(); id[i] = nd.getIdAsInteger();
        }
        if (rm_group_nodes_upload(grp_, int32_t(n), x.data(), y.data(), z.data(), tp.data(), ch.data(), en.data(), rp.data(), xp.data(),
                                  id.data()) != RM_OK) {
            lastError = rm_last_error();
            return false;
        }
        dst_.resize(n + 1); verdict_.resize(n + 1); rssi_.resize(n + 1);
        uploaded_ = sim->nodesVersion();
        return true;
    }
    int kind_;
    rm_group *grp_ = nullptr;
    rm_model_params params_{};
    uint64_t uploaded_ = ~0ull;
    std::vector<int32_t> dst_;
    std::vector<uint8_t> verdict_;
    std::vector<double> rssi_;
};


// ---- packet traces ----------------------------------------------------
// util/PcapExporter.java:47-91: classic pcap, every field written big-endian by DataOutputStream:
// magic 0xa1b2c3d4, version 2.4, thiszone 0, sigfigs 0, snaplen 4096, network 195
// (LINKTYPE_IEEE802_15_4); per packet ts_sec = time / 1e6, ts_usec = time % 1e6 (time in
// microseconds, truncated to int as in the reference), incl_len = orig_len = data length, the bytes.
class PcapExporter {
public:
    ~PcapExporter() { closePcap(); }
    void openPcap(const std::string &pcapFile)
    {
        closePcap();
        out_ = std::fopen(pcapFile.c_str(), "wb");
        if (!out_) throw std::runtime_error("cannot open " + pcapFile);
        writeInt(0xa1b2c3d4u);
        writeShort(0x0002);
        writeShort(0x0004);
        writeInt(0);
        writeInt(0);
        writeInt(4096);
        writeInt(195);
        std::fflush(out_);
    }
    void closePcap()
    {
        if (out_) std::fclose(out_);
        out_ = nullptr;
    }
    bool isOpen() const { return out_ != nullptr; }
    void exportPacketData(int64_t time, const std::vector<uint8_t> &data)
    {
        if (!out_) throw std::runtime_error("pcap file never set"); // the reference opens "radiolog-<millis>.pcap" here
        writeInt(uint32_t(int32_t(time / 1000000)));
        writeInt(uint32_t(int32_t(time % 1000000)));
        writeInt(uint32_t(data.size()));
        writeInt(uint32_t(data.size()));
        if (!data.empty()) std::fwrite(data.data(), 1, data.size(), out_);
        std::fflush(out_);
    }

private:
    void writeInt(uint32_t v)
    {
        const uint8_t b[4] = {uint8_t(v >> 24), uint8_t(v >> 16), uint8_t(v >> 8), uint8_t(v)};
        std::fwrite(b, 1, 4, out_);
    }
    void writeShort(uint16_t v)
    {
        const uint8_t b[2] = {uint8_t(v >> 8), uint8_t(v)};
        std::fwrite(b, 1, 2, out_);
    }
    std::FILE *out_ = nullptr;
};

// util/PcapListener.java:40-58
class PcapListener : public RadioListener {
public:
    explicit PcapListener(const std::string &file) { exporter.openPcap(file); }
    void packetTransmission(RadioPacket &packet) override
    {
        exporter.exportPacketData(packet.getStartTime(), packet.getPacketDataAsBytes());
    }
    PcapExporter exporter;
};

// A pcap file does not say which node sent a frame, so it cannot be replayed.  The compact trace is
// what a replay needs and nothing else: a 16-byte header ("RMTRACE1", record count) and one
// little-endian 32-byte record per transmission, in call order:
//   int64 time_us | int32 source node index | int32 hex length | float64 rf-power | int32 channel | int32 0
// radio-sim_amd/trace.py reads it back and feeds whole ticks of it to rm_batch_run_device.
class TraceListener : public RadioListener {
public:
    explicit TraceListener(const std::string &file) : out_(std::fopen(file.c_str(), "wb"))
    {
        if (!out_) throw std::runtime_error("cannot open " + file);
        writeHeader();
    }
    ~TraceListener() { close(); }
    void packetTransmission(RadioPacket &p) override
    {
        struct Rec {
            int64_t time;
            int32_t src, hexLength;
            double power;
            int32_t channel, zero;
        } r = {p.getStartTime(), int32_t(p.getSource()->index), int32_t(p.getPacketDataAsHex().size()), p.getTransmitPower(),
               int32_t(p.getWirelessChannel()), 0};
        static_assert(sizeof(Rec) == 32, "trace record layout");
        std::fwrite(&r, sizeof(r), 1, out_);
        ++count_;
    }
    void close()
    {
        if (!out_) return;
        std::fseek(out_, 0, SEEK_SET);
        writeHeader();
        std::fclose(out_);
        out_ = nullptr;
    }

private:
    void writeHeader()
    {
        std::fwrite("RMTRACE1", 1, 8, out_);
        std::fwrite(&count_, sizeof(count_), 1, out_);
    }
    std::FILE *out_;
    uint64_t count_ = 0;
};

} // namespace emul8
