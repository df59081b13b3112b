"""oracle/fsm_ref.py -- TEST INFRASTRUCTURE (checker only; nothing under edison_amd/ may import it).

A CPU restatement of the reference firmware's home-automation state machine, `edisonFSM`
(/root/reference/firmware/src/app.c:727-928), written from that source WITHOUT the product's csrc/edison_fsm_core.h: the checker of
`edison_fsm_step` (host) and of the GPU stage behind the output filter (`edison_postproc`, `edison_stream_opts.fsm`).

PARITY UNPINNED: the reference holds no vectors, tests or fixtures for this function (it drives an LED strip on the board); what can be
pinned is its text, which this file follows statement by statement, keeping the firmware's own data structures:

  * the tables `ediLocations[]` / `ediValues[]` (app.c:134-146) with a NULL-name terminator and a `keywordIdx` field that is 0 until the
    EDI_RESET state fills it by NAME from the network's keyword list (app.c:770-784; keyword list = firmware/src/ai/nnom/keywords.txt);
  * the static `loc` / `val` "pointers" (indices into those tables here) that the search loops of HOT and LOC leave where they stop --
    on the matching entry after `break`, on the terminator otherwise (app.c:810-819, 835-843);
  * `hotTimeout += dt/1000` in uint32 arithmetic: integer division, so a dt below 1000 us adds nothing (app.c:805, 830);
  * the ORDER of the two tests in HOT and LOC: first the keyword test, then `hotTimeout > EDI_LOC_TIMEOUT` -- in HOT a location resets
    the counter before the time-out test, in LOC a value found on the very call that times out is overridden by the time-out
    (app.c:807-825, 832-849);
  * `*predMax > TRUE_THRESHOLD`: a float compared with a double constant (app.c:34, 797, 807, 832).

Left out: the LED animations (app.c:737-755, 853-925) and the timer read that produces dt (app.c:757-761; dt is an argument here).
"""

EDI_RESET, EDI_IDLE, EDI_HOT, EDI_LOC, EDI_SET = 0, 1, 2, 3, 4          # app.c:101-108
STATE_NAMES = ["RESET", "IDLE", "HOT", "LOC", "SET"]                       # app.c:109
TRUE_THRESHOLD = 0.5                                                       # app.c:34
EDI_LOC_TIMEOUT = 5000                                                     # app.c:46, milliseconds
EDI_WAKEWORD = "edison"                                                    # app.c:48
# firmware/src/ai/nnom/keywords.txt (the list aiGetKeywordFromIndex answers from)
KEYWORDS = ["edison", "cinema", "bedroom", "office", "livingroom", "kitchen", "on", "off", "_cold", "_noise"]
_U32 = 0xFFFFFFFF


def _locations():
    # app.c:134-141: {name, keywordIdx = 0, colour, ledIdx}; the last entry's NULL name ends every loop over the table
    return [dict(name=n, keywordIdx=0) for n in ("cinema", "bedroom", "office", "livingroom", "kitchen", None)]


def _values():
    # app.c:142-146
    return [dict(name=n, keywordIdx=0) for n in ("off", "on", None)]


class EdisonFsmRef:
    """One instance = the function's statics (app.c:729-733) + `ediState` (app.c:133)."""

    def __init__(self, keywords=None, true_threshold=TRUE_THRESHOLD):
        self.keywords = list(KEYWORDS if keywords is None else keywords)
        self.true_threshold = float(true_threshold)
        self.ediState = EDI_RESET
        self.hotTimeout = 0          # static uint32_t
        self.wakeWordIdx = 0         # static uint8_t, zero-initialised
        self.ediLocations = _locations()
        self.ediValues = _values()
        self.loc = 0                 # static pointers: index into the tables
        self.val = 0
        self.executed = []           # what case EDI_SET prints: (location name, value name) (app.c:855)

    # ---- the function
    def step(self, predMax, predMaxIdx, dt_us):
        """One call of edisonFSM with `dt` microseconds since the last one; returns the state the machine is left in."""
        import numpy as np
        predMax = float(np.float32(predMax))          # float *predMax
        predMaxIdx = int(predMaxIdx) & _U32           # uint32_t *predMaxIdx
        dt = int(dt_us) & _U32                        # uint32_t dt
        nextState = self.ediState
        s = self.ediState
        if s == EDI_RESET:                            # app.c:767-793
            for idx in range(len(self.keywords)):
                if self.keywords[idx] == EDI_WAKEWORD:
                    self.wakeWordIdx = idx & 0xFF
                self.loc = 0
                while self.ediLocations[self.loc]["name"] is not None:
                    if self.keywords[idx] == self.ediLocations[self.loc]["name"]:
                        self.ediLocations[self.loc]["keywordIdx"] = idx
                    self.loc += 1
                self.val = 0
                while self.ediValues[self.val]["name"] is not None:
                    if self.keywords[idx] == self.ediValues[self.val]["name"]:
                        self.ediValues[self.val]["keywordIdx"] = idx
                    self.val += 1
            nextState = EDI_IDLE
        elif s == EDI_IDLE:                           # app.c:795-802
            if predMax > self.true_threshold and predMaxIdx == self.wakeWordIdx:
                self.hotTimeout = 0
                nextState = EDI_HOT
        elif s == EDI_HOT:                            # app.c:803-826
            self.hotTimeout = (self.hotTimeout + dt // 1000) & _U32
            if predMax > self.true_threshold:
                self.loc = 0
                while self.ediLocations[self.loc]["name"] is not None:
                    if self.ediLocations[self.loc]["keywordIdx"] == predMaxIdx:
                        self.hotTimeout = 0
                        nextState = EDI_LOC
                        break
                    self.loc += 1
            if self.hotTimeout > EDI_LOC_TIMEOUT:
                nextState = EDI_IDLE
        elif s == EDI_LOC:                            # app.c:828-850
            self.hotTimeout = (self.hotTimeout + dt // 1000) & _U32
            if predMax > self.true_threshold:
                self.val = 0
                while self.ediValues[self.val]["name"] is not None:
                    if self.ediValues[self.val]["keywordIdx"] == predMaxIdx:
                        nextState = EDI_SET
                        break
                    self.val += 1
            if self.hotTimeout > EDI_LOC_TIMEOUT:
                nextState = EDI_IDLE
        elif s == EDI_SET:                            # app.c:852-873: "set location to required value", print, back to idle
            self.executed.append((self.ediLocations[self.loc]["name"], self.ediValues[self.val]["name"]))
            nextState = EDI_IDLE
        else:                                         # app.c:875-876: Error_Handler()
            raise ValueError("edisonFSM: state %r does not exist" % (s,))
        self.ediState = nextState
        return nextState

    # ---- views for comparison with the product's `edison_fsm` struct (include/edison_hip.h)
    def pending_location_idx(self):
        """keyword index of the entry `loc` points at, -1 on the terminator"""
        e = self.ediLocations[self.loc]
        return -1 if e["name"] is None else int(e["keywordIdx"])

    def pending_value_idx(self):
        e = self.ediValues[self.val]
        return -1 if e["name"] is None else int(e["keywordIdx"])

    def last_command_idx(self):
        if not self.executed:
            return (-1, -1)
        l, v = self.executed[-1]
        return (self.keywords.index(l), self.keywords.index(v))


def walk(pred_max, pred_idx, dt_us, true_threshold=TRUE_THRESHOLD, machine=None):
    """The machine over a stream of (filtered maximum, its class) pairs, dt_us apart: returns (states per call, machine)."""
    m = machine if machine is not None else EdisonFsmRef(true_threshold=true_threshold)
    m.true_threshold = float(true_threshold)
    return [m.step(float(p), int(i), dt_us) for p, i in zip(pred_max, pred_idx)], m
