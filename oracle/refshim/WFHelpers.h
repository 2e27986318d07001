/* refshim: <WFHelpers.h> (SuWidgets, absent) */
#ifndef REFSHIM_WFHELPERS_H
#define REFSHIM_WFHELPERS_H
#include <QString>
#include <QColor>
#include <map>
struct FrequencyBand { qint64 min = 0, max = 0; std::string primary, secondary, footnotes; QColor color; };
class FrequencyAllocationTable {
  std::string name;
public:
  FrequencyAllocationTable() = default;
  FrequencyAllocationTable(std::string const &n) : name(n) {}
  void pushBand(FrequencyBand const &) {}
  std::string const &getName() const { return name; }
};
#endif
