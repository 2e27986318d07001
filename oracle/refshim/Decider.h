/* refshim: <Decider.h> (SuWidgets, absent) [UPSTREAM-RECOLLECTION, SPEC.md section K]: what Tasks/WaveSampler.cpp:315-317
 * calls.  Thresholding is the oracle's sdo_decide. */
#ifndef REFSHIM_DECIDER_H
#define REFSHIM_DECIDER_H
#include <sigutils/types.h>
#include <vector>
typedef uint8_t Symbol;
class Decider {
public:
  enum DecisionMode { ARGUMENT, MODULUS };
private:
  DecisionMode mode = ARGUMENT;
  unsigned bps = 1;
  float min = -(float)PI, max = (float)PI;
public:
  void setDecisionMode(DecisionMode m) { mode = m; }
  void setBps(unsigned b) { bps = b; }
  void setMinimum(float v) { min = v; }
  void setMaximum(float v) { max = v; }
  void decide(const SUCOMPLEX *data, Symbol *symbols, size_t len) const;
};
#endif
