#pragma once
#include <opencv2/opencv.hpp>
