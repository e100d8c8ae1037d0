/*
 * GpuRadioMedium -- the reference-side binding a radio-sim maintainer adds to attach the MI355X
 * engine (libradiomedium_hip.so) behind the existing RadioMedium plug-in contract.
 *
 * Build-owned file (not a copy of any reference source).  It lives in the reference's package so
 * that it can extend AbstractRadioMedium (AbstractRadioMedium.java:35-53) and use the
 * package-visible Simulator / Node / Transciever accessors.  NOT compiled in this repository:
 * there is no JDK in the build image (SURVEY.md section 0.4).
 *
 * Semantics mirrored (paths relative to radio-medium/java/se/sics/emul8/radiomedium/):
 *   transmit(RadioPacket)   UDGMRadioMedium.java:83-117 and siblings: one native call, then the
 *                           heard links come back in node order and are turned into exactly the
 *                           calls the reference makes (generateTransmissionEvents once,
 *                           generateReceptionEvents per heard link, or deliverRadioPacket for the
 *                           constant-loss medium, UDGMConstantLossRadioMedium.java:31).
 *   node state              the reference has no change notification (SURVEY.md section 3.3):
 *                           the node table is re-uploaded when its length or a version stamp
 *                           changed; positions / radio fields are read through the public getters.
 *   threading               transmit() is entered from per-socket reader threads
 *                           (net/JSONClientConnection.java:118-131): one lock per context.
 *   errors                  transmit() returns void and must not throw: failures are logged and
 *                           the packet reaches no receiver (its transmission events are generated all
 *                           the same, as every reference medium does first thing).
 *
 * Batching modes (the reference consumes a tick's events only in emulatorTimeStepDone ->
 * processAllEvents, Simulator.java:155-165, so nothing forces one evaluation per packet):
 *   setTickMode(true)       transmit() only queues the packet.  flush() -- ONE call a maintainer adds at
 *                           the top of Simulator.emulatorTimeStepDone, before `currentTime = stepTime` --
 *                           evaluates the queue in one pass (nTickBegin / nEnqueue / nTickFlushView) and
 *                           makes exactly the per-packet mode's generate*Events calls, in arrival x node
 *                           order.  12-13 us of GPU time per tick of 1000 frames at 100k nodes instead of
 *                           1000 x 22 us.
 *   setDeviceEvents(true)   the events never reach the JVM: the engine keeps packets, heard links and
 *                           every node's radio state on the device (rm_events_*).  processEvents(time) --
 *                           called INSTEAD of processAllEvents(currentTime) -- returns the drain's
 *                           deliveries in the order the reference's ladder queue pops them and turns them
 *                           into Simulator.deliverRadioPacket calls; nodeInfo() serves the per-node fields
 *                           of the time-step message (net/JSONClientConnection.java:331-341).
 */
package se.sics.emul8.radiomedium;

import org.slf4j.Logger;
import org.slf4j.LoggerFactory;

public class GpuRadioMedium extends AbstractRadioMedium {

    private static final Logger log = LoggerFactory.getLogger(GpuRadioMedium.class);

    public static final int MODEL_NULL = 0, MODEL_UDGM = 1, MODEL_UDGM_CONST = 2, MODEL_N2N = 3, MODEL_LOGDIST = 4;
    private static final byte INTERFERED = 1, DELIVERED = 2;

    /** RM_ABI_VERSION of the include/radiomedium_hip.h this shim and its JNI glue were written against */
    private static final int ABI_VERSION = 5;

    static {
        System.loadLibrary("radiomedium_jni"); // integration/jni/rm_jni.c, links libradiomedium_hip.so
        if (nAbiVersion() != ABI_VERSION) {
            throw new UnsatisfiedLinkError("libradiomedium_hip.so has ABI version " + nAbiVersion() + ", this shim needs " + ABI_VERSION);
        }
    }

    /* ---- native side: one-to-one with include/radiomedium_hip.h ---- */
    private static native int nAbiVersion();
    private static native long nCreate(int device);
    private static native void nDestroy(long ctx);
    private static native String nLastError();
    private static native String nGetName(long ctx);
    private static native int nSetModel(long ctx, int kind, int flags, double[] params);
    private static native int nSetN2NMatrix(long ctx, int m, double[] rowMajor);
    private static native int nSeed(long ctx, long seed);
    private static native int nNodesUpload(long ctx, int n, double[] x, double[] y, double[] z, double[] txpower,
            int[] channel, byte[] enabled, double[] rxprob, double[] txprob, int[] intId);
    private static native int nNodeUpdate(long ctx, int node, double x, double y, double z, double txpower, int channel,
            boolean enabled, double rxprob, double txprob);
    private static native int nSetTime(long ctx, long currentTime);
    /** returns the number of heard links (negative = rm error); dst/verdict/rssi are filled in node order */
    private static native int nTransmit(long ctx, int src, long startUs, long hexLength, boolean hasPower, double txpower,
            boolean hasChannel, int channel, int[] dst, byte[] verdict, double[] rssi, double[] sinr, byte[] interference);

    /* tick mode: the queued transmit() calls in one evaluation; the result is read in place from direct buffers
     * over the context's pinned block (rm_tick_flush_view); views = {pktOffset, pktInterference, dst, verdict, rssi} */
    private static native int nTickBegin(long ctx, long tBegin, long tEnd);
    private static native int nEnqueue(long ctx, int src, long startUs, long airUs, double txpower, int channel);
    private static native int nTickRun(long ctx);
    private static native int nTickFlushView(long ctx, java.nio.ByteBuffer[] views, int[] counts /* links, packets */);
    /* reception stage on the device */
    private static native int nEventsEnable(long ctx, int maxPackets, int maxLinks);
    private static native long nEventsNextPacket(long ctx);
    private static native int nEventsProcess(long ctx, long timeUs, java.nio.ByteBuffer[] views /* run packet, run first, run count, dst, rssi */,
            long[] counts /* deliveries, pending packets, number of the oldest pending packet, runs */);
    private static native int nNodeInfo(long ctx, int[] nodes, double[] rssi, int[] receiving, int[] channel);
    /* returns the number of nodes whose node-info changed since it was last reported (< 0: error); the arrays take that many */
    private static native int nNodeInfoChanged(long ctx, int[] nodes, double[] rssi, int[] receiving, int[] channel);

    private final Object lock = new Object();
    private final int kind;
    private boolean tickMode, deviceEvents;
    private final java.util.ArrayList<RadioPacket> queue = new java.util.ArrayList<RadioPacket>();
    private final java.util.ArrayDeque<RadioPacket> inFlight = new java.util.ArrayDeque<RadioPacket>();
    private long firstInFlight;
    private double[] params; // the double fields of rm_model_params, udgm_success_ratio_tx first (nSetModel)
    private int flags;       // RM_LD_* (nSetModel)
    private long ctx;
    private Node[] uploaded;          // the Simulator.getNodes() snapshot the device currently mirrors
    private java.util.IdentityHashMap<Node, Integer> index = new java.util.IdentityHashMap<Node, Integer>();
    private final java.util.ArrayList<Node> changed = new java.util.ArrayList<Node>(); // the dirty list
    private int[] dst = new int[0];
    private byte[] verdict = new byte[0];
    private double[] rssi = new double[0], sinr = new double[0];

    public GpuRadioMedium(int kind, long randomSeed) {
        this.kind = kind;
        this.ctx = nCreate(0);
        if (this.ctx == 0) {
            throw new IllegalStateException("no MI355X radio medium: " + nLastError());
        }
        nSetModel(ctx, kind, 0, null);
        // The engine keeps the generator on the device and consumes exactly the draws the Java loop would.  A run is
        // reproducible only with a seeded Simulator(new Random(seed)) (Simulator.java:87-89) and the same seed here;
        // simulator.getRandom() itself is never advanced by this medium.
        nSeed(ctx, randomSeed);
    }

    /* the reference media's setters (UDGMRadioMedium.java:31-61, N2NRadioMedium.java:11) */
    private void param(int index, double v) {
        synchronized (lock) {
            if (params == null) {
                params = new double[] {1.0, 1.0, 50.0, 100.0, 100.0}; // UDGMRadioMedium.java:18-24, UDGMConstantLossRadioMedium.java:8
            }
            params[index] = v;
            if (nSetModel(ctx, kind, flags, params) != 0) {
                log.error("radio medium: {}", nLastError());
            }
        }
    }
    /* the extension medium (MODEL_LOGDIST; DESIGN.md section 6): every parameter at once, in the order of
     * rm_model_params' double fields after the five of the reference media -- pl0, exponent, d0, sigma, clip,
     * sensitivity, noise, capture, interference floor; the shadowing seed keeps rm_model_defaults' value (nSetModel
     * carries the double fields and the flags only) */
    public void setLogDistance(double pl0Db, double exponent, double d0, double sigmaDb, double clip, double sensitivityDbm,
                               double noiseDbm, double captureDb, double interferenceFloorDbm, boolean sinr) {
        synchronized (lock) {
            params = new double[] {1.0, 1.0, 50.0, 100.0, 100.0, pl0Db, exponent, d0, sigmaDb, clip, sensitivityDbm, noiseDbm,
                                   captureDb, interferenceFloorDbm};
            flags = sinr ? 1 : 0; // RM_LD_SINR
            if (nSetModel(ctx, kind, flags, params) != 0) {
                log.error("radio medium: {}", nLastError());
            }
        }
    }
    public void setSuccessRatioTx(double v) { param(0, v); }   // declared by the reference, never used by it
    public void setSuccessRatioRx(double v) { param(1, v); }
    public void setTransmissionRange(double v) { param(2, v); }
    public void setInterferenceRange(double v) { param(3, v); } // declared by the reference, never read by it
    public void setMatrix(double[][] m) {                       // N2NRadioMedium(double[][]): rows x longest row
        int cols = 0;
        for (double[] row : m) cols = Math.max(cols, row.length);
        int dim = Math.max(m.length, cols);
        double[] flat = new double[dim * dim];
        for (int i = 0; i < m.length; i++) System.arraycopy(m[i], 0, flat, i * dim, m[i].length);
        synchronized (lock) {
            if (nSetN2NMatrix(ctx, dim, flat) != 0) {
                log.error("radio medium: {}", nLastError());
            }
        }
    }

    public void setTickMode(boolean on) { synchronized (lock) { tickMode = on; } }

    public void setDeviceEvents(boolean on) {
        synchronized (lock) {
            if (nEventsEnable(ctx, on ? 1 << 16 : 0, on ? 1 << 21 : 0) != 0) {
                log.error("radio medium: {}", nLastError());
                return;
            }
            deviceEvents = on;
            inFlight.clear();
            firstInFlight = on ? nEventsNextPacket(ctx) : 0;
        }
    }

    @Override
    public String getName() {
        return nGetName(ctx) + " [MI355X]";
    }

    private void syncNodes(Node[] nodes) {
        if (nodes == uploaded) { // copy-on-write array (Simulator.java:274): same array == same node set
            for (Node nd : changed) { // the dirty list: rm_node_update writes these nodes in place on the device
                Integer i = index.get(nd);
                Transciever r = nd.getRadio();
                if (i != null && nNodeUpdate(ctx, i, nd.getPosition().x, nd.getPosition().y, nd.getPosition().z,
                        r.getTransmitPower(), r.getWirelessChannel(), r.isEnabled(), r.getRxProbability(),
                        r.getTxProbability()) != 0) {
                    log.error("node update failed: {}", nLastError());
                }
            }
            changed.clear();
            return;
        }
        changed.clear(); // covered by the snapshot below
        int n = nodes.length;
        double[] x = new double[n], y = new double[n], z = new double[n], tp = new double[n], rp = new double[n], xp = new double[n];
        int[] ch = new int[n], id = new int[n];
        byte[] en = new byte[n];
        index.clear();
        for (int i = 0; i < n; i++) {
            Node nd = nodes[i];
            Transciever r = nd.getRadio();
            x[i] = nd.getPosition().x; y[i] = nd.getPosition().y; z[i] = nd.getPosition().z;
            tp[i] = r.getTransmitPower(); ch[i] = r.getWirelessChannel(); en[i] = (byte) (r.isEnabled() ? 1 : 0);
            rp[i] = r.getRxProbability(); xp[i] = r.getTxProbability(); id[i] = nd.getIdAsInteger();
            index.put(nd, i);
        }
        if (nNodesUpload(ctx, n, x, y, z, tp, ch, en, rp, xp, id) != 0) {
            log.error("node upload failed: {}", nLastError());
        }
        uploaded = nodes;
        if (dst.length < n) {
            dst = new int[n]; verdict = new byte[n]; rssi = new double[n]; sinr = new double[n];
        }
    }

    /** call after node-config-set changed fields of an existing node (no hook exists in the reference,
     *  SimulatorJSONHandler.java:105-143): only this node is written to the device before the next packet */
    public void nodeChanged(Node node) {
        synchronized (lock) {
            changed.add(node);
        }
    }

    /** anything may have changed: the next packet uploads a fresh snapshot */
    public void invalidateNodes() {
        synchronized (lock) {
            uploaded = null;
        }
    }

    @Override
    public void transmit(RadioPacket packet) {
        Simulator sim = this.simulator;
        if (sim == null) {
            log.error("No simulator"); // NullRadioMedium.java:49-53
            return;
        }
        Node[] nodes = sim.getNodes();
        if (nodes == null) {
            return;
        }
        synchronized (lock) {
            if (tickMode) { // evaluated in flush(), with everything else that is sent in this tick
                queue.add(packet);
                return;
            }
            syncNodes(nodes);
            Integer src = index.get(packet.getSource());
            if (src == null) {
                log.error("source node not registered");
                return;
            }
            nSetTime(ctx, sim.getTime());
            byte[] interference = new byte[1];
            int heard = nTransmit(ctx, src, packet.getStartTime(), packet.getPacketDataAsHex() == null ? 0
                    : packet.getPacketDataAsHex().length(), true, packet.getTransmitPower(), true,
                    packet.getWirelessChannel(), dst, verdict, rssi, sinr, interference);
            if (deviceEvents) { // the engine queued the packet's events itself -- if it numbered the packet
                if (heard < 0) log.error("radio medium: {}", nLastError());
                if (nEventsNextPacket(ctx) == firstInFlight + inFlight.size() + 1) inFlight.add(packet);
                return;
            }
            if (kind != MODEL_UDGM_CONST) {
                sim.generateTransmissionEvents(packet); // UDGMRadioMedium.java:97 -- whatever the native call said
            }
            if (heard < 0) {
                log.error("radio medium: {}", nLastError());
                return;
            }
            for (int i = 0; i < heard; i++) { // node order, as the reference's loop (:99)
                Node node = nodes[dst[i]];
                if (kind == MODEL_UDGM_CONST) {
                    sim.deliverRadioPacket(packet, node, rssi[i]); // immediate delivery, no events
                } else {
                    sim.generateReceptionEvents(packet, node, rssi[i], verdict[i] == DELIVERED);
                }
            }
        }
    }

    /** Tick mode: call at the top of Simulator.emulatorTimeStepDone, before `this.currentTime = stepTime`
     *  (Simulator.java:156): the event times of the queued packets are max(start, OLD currentTime), :323-326. */
    public void flush() {
        Simulator sim = this.simulator;
        synchronized (lock) {
            if (queue.isEmpty() || sim == null) {
                queue.clear();
                return;
            }
            Node[] nodes = sim.getNodes();
            syncNodes(nodes);
            nSetTime(ctx, sim.getTime());
            int rc = nTickBegin(ctx, sim.getTime(), sim.getTime());
            // the packets the engine actually took, in the order it numbers them: results are indexed by THIS list
            // (a packet whose source is not in the node table is not enqueued and must not shift the others)
            java.util.ArrayList<RadioPacket> enqueued = new java.util.ArrayList<>(queue.size());
            for (RadioPacket p : queue) {
                Integer src = index.get(p.getSource());
                if (rc == 0 && src != null) {
                    rc = nEnqueue(ctx, src, p.getStartTime(), p.getPacketAirTime(), p.getTransmitPower(), p.getWirelessChannel());
                    if (rc == 0) enqueued.add(p);
                }
            }
            if (deviceEvents) {
                long before = nEventsNextPacket(ctx);
                if (rc == 0) rc = nTickRun(ctx);
                // in flight = numbered by the engine: only after a successful run, and only if the numbers agree
                if (rc != 0) log.error("radio medium: {}", nLastError());
                else if (nEventsNextPacket(ctx) - before != enqueued.size() || before != firstInFlight + inFlight.size())
                    log.error("radio medium: packet numbers out of step with the engine; tick dropped");
                else inFlight.addAll(enqueued);
                queue.clear();
                return;
            }
            java.nio.ByteBuffer[] v = new java.nio.ByteBuffer[6];
            int[] counts = new int[2];
            if (rc == 0) rc = nTickFlushView(ctx, v, counts);
            if (rc != 0) log.error("radio medium: {}", nLastError());
            java.nio.IntBuffer off = rc == 0 ? v[0].order(java.nio.ByteOrder.nativeOrder()).asIntBuffer() : null;
            java.nio.IntBuffer d = rc == 0 ? v[2].order(java.nio.ByteOrder.nativeOrder()).asIntBuffer() : null;
            // ABI version 5: the links' rssi (v[4]) -- or, for the reference's own media, one value per PACKET (v[5]): a heard link's
            // rssi is packet.getTransmitPower() there (UDGMRadioMedium.java:95), and it crosses PCIe once per packet
            boolean perPacket = rc == 0 && v[4].capacity() == 0 && counts[0] > 0;
            java.nio.DoubleBuffer r = rc == 0 ? v[perPacket ? 5 : 4].order(java.nio.ByteOrder.nativeOrder()).asDoubleBuffer() : null;
            int k = 0; // index into the engine's packets of this tick
            for (RadioPacket p : queue) { // arrival order, then node order: the per-packet calls
                if (kind != MODEL_UDGM_CONST) sim.generateTransmissionEvents(p); // whatever the native call said
                if (rc != 0 || k >= enqueued.size() || enqueued.get(k) != p) continue; // not evaluated: no receivers
                for (int i = off.get(k); i < off.get(k + 1); i++) {
                    Node node = nodes[d.get(i)];
                    double rssi = r.get(perPacket ? k : i);
                    if (kind == MODEL_UDGM_CONST) sim.deliverRadioPacket(p, node, rssi);
                    else sim.generateReceptionEvents(p, node, rssi, v[3].get(i) == DELIVERED);
                }
                k++;
            }
            queue.clear();
        }
    }

    /** Device events: call INSTEAD of processAllEvents(currentTime) in Simulator.emulatorTimeStepDone (:161). */
    public void processEvents(long time) {
        Simulator sim = this.simulator;
        synchronized (lock) {
            if (!deviceEvents || sim == null) return;
            java.nio.ByteBuffer[] v = new java.nio.ByteBuffer[5];
            long[] counts = new long[4]; // deliveries, pending packets, number of the oldest pending packet, runs
            if (nEventsProcess(ctx, time, v, counts) != 0) {
                log.error("radio medium: {}", nLastError());
                return;
            }
            Node[] nodes = sim.getNodes();
            // the deliveries of one packet are adjacent in the queue's pop order: its number comes once per run
            java.nio.LongBuffer runPacket = v[0].order(java.nio.ByteOrder.nativeOrder()).asLongBuffer();
            java.nio.IntBuffer runFirst = v[1].order(java.nio.ByteOrder.nativeOrder()).asIntBuffer();
            java.nio.IntBuffer runCount = v[2].order(java.nio.ByteOrder.nativeOrder()).asIntBuffer();
            java.nio.IntBuffer d = v[3].order(java.nio.ByteOrder.nativeOrder()).asIntBuffer();
            java.nio.DoubleBuffer r = v[4].order(java.nio.ByteOrder.nativeOrder()).asDoubleBuffer();
            RadioPacket[] live = inFlight.toArray(new RadioPacket[0]);
            for (int run = 0; run < counts[3]; run++) {
                long k = runPacket.get(run) - firstInFlight;
                int first = runFirst.get(run), end = first + runCount.get(run);
                if (k < 0 || k >= live.length || first < 0 || end > counts[0]) {
                    log.error("radio medium: a delivery names a packet the host does not hold");
                    continue;
                }
                for (int i = first; i < end; i++) { // ReceptionEvent.java:41-44, in the queue's pop order
                    if (d.get(i) < 0 || d.get(i) >= nodes.length) {
                        log.error("radio medium: a delivery names a node the host does not hold");
                        continue;
                    }
                    sim.deliverRadioPacket(live[(int) k], nodes[d.get(i)], r.get(i));
                }
            }
            long oldest = counts[2]; // every packet below the oldest one still queued has fired its last event
            while (firstInFlight < oldest && !inFlight.isEmpty()) {
                inFlight.poll();
                firstInFlight++;
            }
        }
    }

    /** the node-info fields of a time-step message for `nodes` (indices into Simulator.getNodes()) */
    public boolean nodeInfo(int[] nodes, double[] rssi, int[] receiving, int[] channel) {
        synchronized (lock) {
            return nNodeInfo(ctx, nodes, rssi, receiving, channel) == 0;
        }
    }

    /**
     * The nodes whose (rssi, receiving state, channel) differ from what this call reported for them last -- every node the
     * first time.  JSONClientConnection.emulateToTime writes every node's fields into every time-step message; a connection
     * that keeps the text it sent last rewrites only these.  The arrays must hold one entry per node; returns how many were
     * filled, or -1.
     */
    public int nodeInfoChanged(int[] nodes, double[] rssi, int[] receiving, int[] channel) {
        synchronized (lock) {
            if (simulator != null) syncNodes(simulator.getNodes());
            return nNodeInfoChanged(ctx, nodes, rssi, receiving, channel);
        }
    }

    public void close() {
        synchronized (lock) {
            if (ctx != 0) {
                nDestroy(ctx);
                ctx = 0;
            }
        }
    }
}
