// TEST-ONLY (see tests/stubs/README.md): the host project's System.h pulls in the standard headers the mapper relies on.
#pragma once
#include <list>
#include <mutex>
#include <set>
#include <string>
#include <vector>
