/* the reference splits its public API over several headers; libocrshim keeps it in ocr.h */
#include "ocr.h"
