/* Source-compatibility forwarder: the reference ships this declaration set as include/polycap-progress-monitor.h;
 * in this build every public declaration lives in polycap.h. */
#ifndef POLYCAP_FWD_PROGRESS_MONITOR_H
#define POLYCAP_FWD_PROGRESS_MONITOR_H
#include "polycap.h"
#endif
