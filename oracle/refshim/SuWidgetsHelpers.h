/* refshim: <SuWidgetsHelpers.h> (SuWidgets, absent).  Only what the compiled TUs mention. */
#ifndef REFSHIM_SUWIDGETSHELPERS_H
#define REFSHIM_SUWIDGETSHELPERS_H
#include <QString>
#include <QColor>
#define SCAST(type, value) static_cast<type>(value)
struct BookmarkInfo {
  QString name;
  qint64 frequency = 0;
  QColor color;
  qint32 lowFreqCut = 0, highFreqCut = 0;
  QString modulation;
};
class SuWidgetsHelpers {
public:
  static QString formatQuantity(qreal value, int precision, QString const &units = "", bool sign = false);
  static QString formatQuantity(qreal value, QString const &units = "");
};
#endif
